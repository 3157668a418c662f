#!/bin/bash
# round 5: 4-wave workgroups (64x80 / 64x64 wave tiles) on the 128x160 / 128x128 block tiles vs the 8-wave ones: isolated (gemm_ab, cold operands), then in the step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
P=aozora_sdxl_training_amd
hipcc -O2 --offload-arch=gfx950 -o tools/gemm_ab tools/gemm_ab.cpp -ldl || exit 1
{
echo "## NT: base (8 waves of 32x80) vs w4nt (4 waves of 64x80), 2-stage (backward) and 3-stage (LDS_EXCLUSIVE, forward)"
timeout -k 10 300 tools/gemm_ab $P/lib_exp_base.so $P/lib_exp_w4nt.so -- nt:4096:1280:1280 nt:4096:1280:3840 nt:4096:1280:5120 nt:16384:640:640 opt:LDS_EXCLUSIVE=1 nt:4096:1280:1280 nt:4096:1280:5120 nt:16384:640:640
echo "## TN / conv weight gradients: base (8 waves of 32x64) vs w4tn (4 waves of 64x64)"
timeout -k 10 300 tools/gemm_ab $P/lib_exp_base.so $P/lib_exp_w4tn.so -- tn:1280:1280:4096:0:b tn:640:640:16384:0:b tn:10240:1280:4096:0:b tn:3840:1280:4096 tn:1280:5120:4096:0:b cw:4:128:128:320:320 cw:4:32:32:1280:1280
} > $O/r05_w4_lab.txt 2>&1
cat $O/r05_w4_lab.txt | grep -v "^$" | tail -60
{
for r in 1 2 3; do
  for v in base w4nt w4tn; do AZ_LIB=$P/lib_exp_$v.so timeout -k 10 300 python3 tools/policy_time.py "" 2>&1 | grep -v amdgpu | sed "s|^|$v  |"; done
done
} > $O/r05_w4_step.txt 2>&1
cat $O/r05_w4_step.txt
