#!/bin/bash
# Interleaved same-box A/B of two ExecPolicy / option strings: bash tools/pol_ab.sh "<a>" "<b>" [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for r in $(seq 1 ${3:-3}); do
timeout -k 10 300 python tools/policy_time.py "$1" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/policy_time.py "$2" 2>&1 | grep -v amdgpu.ids
done
} > $O/pol_ab.txt 2>&1
cat $O/pol_ab.txt
