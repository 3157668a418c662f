#!/bin/bash
# Evidence run of a round (EVID_TAG=rNN_x) (on the GPU box): bench line, rocprofv3 kernel stats (two-stream and TRUE one-stream), per-shape event
# breakdown, PMC passes (FETCH / WRITE / SQ set) over the step's dominant products of every class.  Outputs under gpurun_out/${TAG}_*;
# the summaries are copied into profiles/ afterwards (see profiles/README.md).
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; TAG=${EVID_TAG:-r05}; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 900 python3 $R/bench.py --steps 10 --warmup 2 --profile-out $O/${TAG}_event_breakdown.json > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err || { echo "bench failed"; tail -5 $O/${TAG}_bench.err; exit 1; }
echo "bench done"; head -c 600 $O/${TAG}_bench_line.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -o stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --through-trainer 0 > $O/prof_${TAG}.log 2>&1 || { echo "stats failed"; tail -5 $O/prof_${TAG}.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_serial -o stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc --through-trainer 0 --serial > $O/prof_${TAG}_serial.log 2>&1 || { echo "serial stats failed"; tail -5 $O/prof_${TAG}_serial.log; exit 1; }
rm -f $O/prof_${TAG}*/*kernel_trace.csv $O/prof_${TAG}*/*/*kernel_trace.csv
echo "stats done"
cd $R
AZ_SHAPES=1 AZ_TOP=130 timeout -k 10 300 python3 tools/class_breakdown.py > $O/${TAG}_shape_breakdown.txt 2>&1 || { echo "breakdown failed"; exit 1; }
export PMC_MANIFEST=$O/${TAG}_pmc_manifest.json
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES"; do
  T=${P%%:*}; C=${P#*:}
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${TAG}_pass_$T -o p -- python3 tools/pmc_target.py > $O/pmc_${TAG}_pass_$T.log 2>&1 || { echo "PMC pass $T failed"; tail -3 $O/pmc_${TAG}_pass_$T.log; exit 1; }
  python3 tools/pmc_collect.py $O/pmc_${TAG}_pass_$T $PMC_MANIFEST $O/${TAG}_pmc_$T.json || exit 1
  rm -rf $O/pmc_${TAG}_pass_$T
  echo "pmc pass $T done"
done

# stream timeline of 8 micro-steps (busy / idle per queue, per phase of the last micro-step)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_${TAG} -o t -- python3 $R/tools/trace_target.py > $O/trace_${TAG}.log 2>&1 || { echo "trace failed"; tail -5 $O/trace_${TAG}.log; exit 1; }
python3 $R/tools/trace_gaps.py $(ls $O/trace_${TAG}/*/*kernel_trace.csv $O/trace_${TAG}/*kernel_trace.csv 2>/dev/null | head -1) > $O/${TAG}_stream_gaps.txt 2>&1 || echo "gap analysis failed"
rm -rf $O/trace_${TAG}
echo "trace done"
echo "evidence done"
