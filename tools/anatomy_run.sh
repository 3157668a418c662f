#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=aozora_sdxl_training_amd/lib_exp_anatomy.so
{
echo "== 4096x1280x1280, 128x160 2-stage (backward form)"; timeout -k 5 60 tools/gemm_anatomy $L 4096 1280 1280
echo "== 4096x1280x1280, 128x160 3-stage (exclusive forward form)"; timeout -k 5 60 tools/gemm_anatomy $L 4096 1280 1280 excl
echo "== 4096x1280x5120, 128x160 3-stage"; timeout -k 5 60 tools/gemm_anatomy $L 4096 1280 5120 excl
echo "== 4096x5120x1280, 8-wave 256x320"; timeout -k 5 60 tools/gemm_anatomy $L 4096 5120 1280
echo "== 4096x10240x1280, 8-wave 256x320"; timeout -k 5 60 tools/gemm_anatomy $L 4096 10240 1280
echo "== 4096x3840x1280, 8-wave 256x256"; timeout -k 5 60 tools/gemm_anatomy $L 4096 3840 1280
echo "== 4096x1280x10240, 128x160 2-stage"; timeout -k 5 60 tools/gemm_anatomy $L 4096 1280 10240
echo "== 4096x4096x4096, 8-wave 256x256"; timeout -k 5 60 tools/gemm_anatomy $L 4096 4096 4096
echo "== 4096x4096x4096, 16-wave 256x256 (round 2)"; timeout -k 5 60 tools/gemm_anatomy $L 4096 4096 4096 tile:256:256:0
} > $O/anatomy.txt 2>&1
cat $O/anatomy.txt
