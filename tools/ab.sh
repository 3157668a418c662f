# same-box A/B of one environment toggle: bash tools/ab.sh VAR A B [repeats]  (alternates, prints the two-stream micro-step times)
cd $GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do for X in $A $B; do echo -n "$V=$X: "; env $V=$X timeout -k 10 200 python3 tools/chain_time.py 2>&1 | grep "2 streams" || exit 1; done; done
