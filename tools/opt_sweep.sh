#!/bin/bash
# Interleaved same-box sweep of option strings: bash tools/opt_sweep.sh rounds "<opts1>" "<opts2>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
N=$1; shift
{
for r in $(seq 1 $N); do
  for o in "$@"; do timeout -k 10 300 python tools/policy_time.py "$o" 2>&1 | grep -v amdgpu.ids; done
done
} > $O/opt_sweep.txt 2>&1
cat $O/opt_sweep.txt
