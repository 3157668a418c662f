#!/bin/bash
# kernel + memory-copy trace of one iteration pair under the N > 1 schedule at one rank (tools/iter_timeline.py EXCHANGE=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export ROUNDS=1 EXCHANGE=${EXCHANGE:-1}
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_exch -o t -- python3 tools/iter_timeline.py > $O/trace_exch.log 2>&1 || { echo "trace failed"; tail -5 $O/trace_exch.log; exit 1; }
python3 tools/exch_trace.py $(ls $O/trace_exch/*/*kernel_trace.csv $O/trace_exch/*kernel_trace.csv 2>/dev/null | head -1) $(ls $O/trace_exch/*/*memory_copy_trace.csv $O/trace_exch/*memory_copy_trace.csv 2>/dev/null | head -1) > $O/exch_trace_${EXCHANGE}.txt 2>&1
rm -rf $O/trace_exch
cat $O/exch_trace_${EXCHANGE}.txt
