#!/bin/bash
# Same-box A/B of bench.py between a worktree of an older commit (_ab_old/, built there) and this tree.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
F="--steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-live-pmc --through-trainer 0"
{
for r in 1 2; do
  (cd $R/_ab_old && timeout -k 10 400 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'])")
  (cd $R && timeout -k 10 400 python bench.py $F --other-configs none 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])")
done
} > $O/bench_ab.txt 2>&1
cat $O/bench_ab.txt
