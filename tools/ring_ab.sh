#!/bin/bash
# A/B of the ring (>= 3 stage) loop with an immediate counted wait against the previous build, plus the loop anatomy.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
NEW=aozora_sdxl_training_amd/libaozora_hip.so; OLD=aozora_sdxl_training_amd/lib_exp_old.so; AN=aozora_sdxl_training_amd/lib_exp_anatomy.so
{
echo "### forward form (LDS_EXCLUSIVE: 3-stage 128x160)"
timeout -k 5 200 tools/gemm_ab $OLD $NEW -- opt:LDS_EXCLUSIVE=1 nt:4096:1280:1280 nt:4096:1280:3840 nt:4096:1280:5120 nt:4096:640:640 nt:16384:640:640 nt:16384:640:2560
echo "### backward form (2-stage)"
timeout -k 5 200 tools/gemm_ab $OLD $NEW -- nt:4096:1280:1280 nt:4096:1280:5120 nt:4096:1280:10240 nt:16384:640:640 tn:1280:1280:4096:0:b tn:10240:1280:4096:0:b cf:4:32:32:1280:1280 cd:4:32:32:1280:1280
echo "### anatomy"
for a in "" excl "opt:GEMM_ABLATE=2" "excl opt:GEMM_ABLATE=2" "excl opt:GEMM_ABLATE=1"; do
  echo "== 4096x1280x1280 $a"; timeout -k 5 60 tools/gemm_anatomy $AN 4096 1280 1280 $a | grep -E "^product|k-loop:"
done
} > $O/ring_ab.txt 2>&1
cat $O/ring_ab.txt
