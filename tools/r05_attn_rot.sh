#!/bin/bash
# round 5: attention workgroup order x rotated tile order, isolated (tools/attn_lab)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc -O2 --offload-arch=gfx950 -o tools/attn_lab tools/attn_lab.cpp -ldl || exit 1
timeout -k 10 500 tools/attn_lab aozora_sdxl_training_amd/libaozora_hip.so -- rounds:9 var:ATTN_XCD=0,ATTN_ROT=0 var:ATTN_XCD=7,ATTN_ROT=0 var:ATTN_XCD=7,ATTN_ROT=7 \
   var:ATTN_XCD=0,ATTN_ROT=7 var:ATTN_XCD=15,ATTN_ROT=0 var:ATTN_XCD=15,ATTN_ROT=7 var:ATTN_XCD=7,ATTN_ROT=2 var:ATTN_XCD=7,ATTN_ROT=4 \
   shape:4:20:1024:1024 shape:4:10:4096:4096 > $O/r05_attn_rot_lab.txt 2>&1 || { tail -30 $O/r05_attn_rot_lab.txt; exit 1; }
grep -E "^shape|fwd |rel-fro" $O/r05_attn_rot_lab.txt
