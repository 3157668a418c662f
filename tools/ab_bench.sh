# same-box A/B of one environment toggle on the whole bench iteration: bash tools/ab_bench.sh VAR A B [repeats]
cd $GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do for X in $A $B; do echo -n "$V=$X: "; env $V=$X timeout -k 10 300 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' || exit 1; done; done
