#!/bin/bash
# Stream timeline only (the last part of r03_evidence.sh): rocprofv3 kernel trace of tools/trace_target.py -> gpurun_out/${TAG}_stream_gaps.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${EVID_TAG:-r03x}; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_${TAG} -o t -- python3 $R/tools/trace_target.py > $O/trace_${TAG}.log 2>&1 || { echo "trace failed"; tail -5 $O/trace_${TAG}.log; exit 1; }
GAP_DUMP="${GAP_DUMP:-}" python3 $R/tools/trace_gaps.py $(ls $O/trace_${TAG}/*/*kernel_trace.csv $O/trace_${TAG}/*kernel_trace.csv 2>/dev/null | head -1) > $O/${TAG}_stream_gaps.txt 2>&1 || echo "gap analysis failed"
rm -rf $O/trace_${TAG}
tail -${TAILN:-40} $O/${TAG}_stream_gaps.txt
