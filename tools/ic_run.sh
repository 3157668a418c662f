#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=aozora_sdxl_training_amd/libaozora_hip.so
{
for cfg in "16 8" "4 8" "64 8" "16 2"; do
for shape in "4096 1280 1280" "4096 5120 1280"; do
  echo "== NT $shape  touch workgroups / group: $cfg"; timeout -k 5 120 tools/ic_prefetch_test $L $shape $cfg
done; done
} > $O/ic_prefetch.txt 2>&1
cat $O/ic_prefetch.txt
