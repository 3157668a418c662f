#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
L=$R/aozora_sdxl_training_amd/libaozora_hip.so
cd $R
SH="nt:4096:1280:10240 nt:4096:1280:5120 nt:4096:1280:3840 nt:4096:1280:1280 nt:16384:640:5120 nt:16384:640:2560"
timeout -k 10 300 tools/gemm_ab $L -- $SH opt:NT_SPLIT_MINK=1280 opt:NT_SPLIT_BIG=3 $SH opt:NT_SPLIT_BIG=4 $SH opt:GEMM8=0 $SH opt:GEMM8=1 opt:NT_SPLIT_BIG=0 nt:4096:5120:64 nt:4096:5120:128 nt:4096:5120:640 nt:4096:5120:2560 nt:4096:5120:5120 > $O/gemm8_ab2.txt 2>&1
echo "rc=$?"; cat $O/gemm8_ab2.txt
