#!/bin/bash
# Operands from HBM (rotating sets > 600 MB), from the Infinity Cache (sets of ~100 MB in total) and from L2 (one set).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=aozora_sdxl_training_amd/lib_exp_anatomy.so
run() { echo "== $*"; timeout -k 5 60 tools/gemm_anatomy $L "$@" | grep -E "^product|k-loop:"; }
{
for st in "" excl; do
  for sets in 25 8 4 2 1; do run 4096 1280 1280 $st sets:$sets; done
  for sets in 10 3 1; do run 4096 1280 5120 $st sets:$sets; done
done
} > $O/anatomy_ic.txt 2>&1
grep -E "==|k-loop|period" $O/anatomy_ic.txt | sed 's/(HIP events.*//'
