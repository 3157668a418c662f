#!/bin/bash
# 8-wave ping-pong tile vs the 16-wave 256x256 tile and the heuristic, same process, cold operands (tools/gemm_ab).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
L=$R/aozora_sdxl_training_amd/libaozora_hip.so
cd $R
SH="nt:4096:10240:1280 nt:4096:5120:1280 nt:4096:3840:1280 nt:16384:5120:640 nt:16384:2560:640 nt:16384:1920:640 nt:4096:4096:4096 nt:8192:8192:8192"
timeout -k 10 300 tools/gemm_ab $L -- opt:GEMM8=0 $SH gg:4096:5120:1280 gg:16384:2560:640 opt:GEMM8=1 $SH gg:4096:5120:1280 gg:16384:2560:640 tile:256:256:8 $SH tile:256:320:8 $SH tile:0:0:0 nt:4096:1280:1280 nt:4096:1280:10240 > $O/gemm8_ab.txt 2>&1
echo "rc=$?"; cat $O/gemm8_ab.txt
