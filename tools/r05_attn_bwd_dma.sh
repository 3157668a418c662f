#!/bin/bash
# round 5: LDS-DMA forms of the attention backward bodies (ATTN_PIPE bit 3) -- tests, isolated A/B (tools/attn_lab), two-stream micro-step A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc -O2 --offload-arch=gfx950 -o tools/attn_lab tools/attn_lab.cpp -ldl 2>/dev/null || exit 1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/r05_bwd_dma_tests.txt 2>&1 || { tail -40 $O/r05_bwd_dma_tests.txt; exit 1; }
tail -2 $O/r05_bwd_dma_tests.txt
timeout -k 10 300 tools/attn_lab aozora_sdxl_training_amd/libaozora_hip.so -- rounds:9 var:ATTN_PIPE=7 var:ATTN_PIPE=15 var:ATTN_PIPE=13 shape:4:20:1024:1024 shape:4:10:4096:4096 > $O/r05_bwd_dma_lab.txt 2>&1 || { tail -30 $O/r05_bwd_dma_lab.txt; exit 1; }
grep -E "^shape|fwd |rel-fro" $O/r05_bwd_dma_lab.txt
bash tools/pol_ab.sh "ATTN_PIPE=7" "ATTN_PIPE=15" 3
cp $O/pol_ab.txt $O/r05_bwd_dma_step.txt
