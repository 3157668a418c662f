# same-box A/B of two builds of the library: bash tools/ab_lib.sh <libA.so> <libB.so> [repeats]  (paths relative to the package dir)
cd $GRAFT_REPO_ROOT
P=aozora_sdxl_training_amd; A=$1; B=$2; R=${3:-2}
cp $P/libaozora_hip.so $P/lib_saved.so
for i in $(seq $R); do for X in $A $B; do cp $P/$X $P/libaozora_hip.so; echo -n "$X: "; timeout -k 10 200 python3 tools/chain_time.py 2>&1 | grep "ms" | tr '\n' ' '; echo; done; done
cp $P/lib_saved.so $P/libaozora_hip.so
