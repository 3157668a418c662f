#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for r in 1 2 3; do
timeout -k 10 300 python tools/policy_time.py "fork_events=0" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/policy_time.py "fork_events=1" 2>&1 | grep -v amdgpu.ids
done
} > $O/fork_ab.txt 2>&1
cat $O/fork_ab.txt
