#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for r in 1 2 3; do
AZ_LIB=aozora_sdxl_training_amd/libaozora_hip.so timeout -k 10 300 python tools/policy_time.py "$1" 2>&1 | grep -v amdgpu.ids
done
} > $O/fork_ab.txt 2>&1
cat $O/fork_ab.txt
