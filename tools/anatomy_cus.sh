#!/bin/bash
# What bounds the k-loop of the 128x160 tile: busy CUs (M = 128 * workgroups / 8), warm vs cold operands, DMA only / MFMA only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=aozora_sdxl_training_amd/lib_exp_anatomy.so
run() { echo "== $*"; timeout -k 5 60 tools/gemm_anatomy $L "$@" | grep -E "^product|k-loop:"; }
{
for M in 128 4096; do
  for st in "" excl; do
    run $M 1280 1280 $st
    run $M 1280 1280 $st sets:1
    run $M 1280 1280 $st opt:GEMM_ABLATE=1
    run $M 1280 1280 $st opt:GEMM_ABLATE=1 sets:1
    run $M 1280 1280 $st opt:GEMM_ABLATE=2
  done
done
} > $O/anatomy_cus2.txt 2>&1
cat $O/anatomy_cus2.txt
