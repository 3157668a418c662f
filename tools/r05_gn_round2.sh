#!/bin/bash
# round 5: does the rounding order of GroupNorm + SiLU move the gradient-norm distance to the fp32 oracle?  (cfg1, 512^2)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in product r2_1 r2_3 r2_7; do
  if [ $v = product ]; then L=; else L=aozora_sdxl_training_amd/lib_exp_$v.so; fi
  echo "=== $v" >> $O/r05_gn_round2.txt
  AZ_LIB=$L timeout -k 10 400 python tools/fp32_gap.py 2>&1 | grep -v amdgpu | head -34 >> $O/r05_gn_round2.txt || exit 1
done
grep -E "^===|global|up_blocks.2.res|R.conv" $O/r05_gn_round2.txt
