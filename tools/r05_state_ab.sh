#!/bin/bash
# Same-box A/B of the headline run with the Raven moments resident in HBM (default) against the reference's residency (pinned host
# memory, 20.5 GB over the host link per optimizer step): bash tools/r05_state_ab.sh  ->  gpurun_out/r05_state_ab.txt
set -e
O=gpurun_out/r05_state_ab.txt; : > $O
for round in 1 2; do
  for mode in "" "--state-on-host"; do
    echo "== round $round ${mode:-resident}" >> $O
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-pmc --through-trainer 0 --other-configs none $mode 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4f it/s  ms_per_step %.1f' % (d['value'], d['ms_per_step'])); print({k: round(v['ms'],2) for k,v in d['exchange']['per_rank'][0].items() if isinstance(v,dict) and 'ms' in v})" >> $O
  done
done
cat $O
