#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
OLD=aozora_sdxl_training_amd/lib_exp_old.so
{
for r in 1 2; do
AZ_LIB=$OLD timeout -k 10 300 python tools/policy_time.py "" 2>&1 | grep -v amdgpu.ids | sed 's/^/old  /'
timeout -k 10 300 python tools/policy_time.py "" 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
timeout -k 10 300 python tools/policy_time.py "TILE_POLICY=5" 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
done
} > $O/ring_step_ab.txt 2>&1
cat $O/ring_step_ab.txt
