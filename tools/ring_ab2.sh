#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
NEW=aozora_sdxl_training_amd/libaozora_hip.so; OLD=aozora_sdxl_training_amd/lib_exp_old.so
{
for tl in 128:160:8 128:160:40 128:160:56 128:160:24 128:160:72; do
echo "### tile $tl"
timeout -k 5 200 tools/gemm_ab $OLD $NEW -- tile:$tl nt:4096:1280:1280 nt:4096:1280:3840 nt:4096:1280:5120 nt:4096:1280:10240 nt:16384:640:640 nt:16384:640:2560
done
} > $O/ring_ab2.txt 2>&1
cat $O/ring_ab2.txt
