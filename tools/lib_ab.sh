#!/bin/bash
# Two-stream micro-step with the previous build of the library (lib_exp_prev.so) and this one, interleaved on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for r in 1 2 3; do
AZ_LIB=aozora_sdxl_training_amd/lib_exp_prev.so timeout -k 10 300 python tools/policy_time.py "$1" 2>&1 | grep -v amdgpu.ids | sed 's/^/prev /'
timeout -k 10 300 python tools/policy_time.py "$1" 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
done
} > $O/lib_ab.txt 2>&1
cat $O/lib_ab.txt
