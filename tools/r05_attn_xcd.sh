#!/bin/bash
# round 5: attention workgroup order (option ATTN_XCD) -- tests, isolated A/B (tools/attn_lab), two-stream micro-step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
set -o pipefail
hipcc -O2 --offload-arch=gfx950 -o tools/attn_lab tools/attn_lab.cpp -ldl || exit 1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/r05_attn_tests.txt 2>&1 || { tail -30 $O/r05_attn_tests.txt; exit 1; }
tail -3 $O/r05_attn_tests.txt
timeout -k 10 300 tools/attn_lab aozora_sdxl_training_amd/libaozora_hip.so -- rounds:9 var:ATTN_XCD=0 var:ATTN_XCD=7 var:ATTN_XCD=1 var:ATTN_XCD=2 \
   shape:4:20:1024:1024 shape:4:10:4096:4096 shape:4:20:1024:77 shape:4:10:4096:77 > $O/r05_attn_xcd_lab.txt 2>&1 || { tail -30 $O/r05_attn_xcd_lab.txt; exit 1; }
cat $O/r05_attn_xcd_lab.txt
bash tools/pol_ab.sh "ATTN_XCD=0" "ATTN_XCD=7" 3
cp $O/pol_ab.txt $O/r05_attn_xcd_step.txt
