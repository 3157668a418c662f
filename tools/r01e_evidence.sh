#!/bin/bash
# Round-1 evidence run (on the GPU box): bench line, rocprofv3 kernel stats, per-shape breakdown, PMC passes on the dominant class.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py --steps 5 --warmup 1 --profile-out $O/r01_e_event_breakdown.json > $O/r01_e_bench_line.json 2> $O/r01_e_bench.err
echo "bench done"; cat $O/r01_e_bench_line.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01e -o stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_r01e.log 2>&1
echo "stats done"
rm -f $O/prof_r01e/*kernel_trace.csv $O/prof_r01e/*/*kernel_trace.csv
cd $R
AZ_SHAPES=1 AZ_TOP=120 timeout -k 10 300 python3 tools/class_breakdown.py > $O/r01_e_shape_breakdown.txt 2>&1
for S in "4096 10240 1280" "4096 1280 10240" "4096 1280 1280" "4096 1280 5120" "4096 5120 1280" "4096 3840 1280" "4096 1280 3840" \
         "16384 640 640" "16384 5120 640" "16384 640 5120" "16384 640 2560" "16384 2560 640" "16384 1920 640" "16384 640 1920"; do
  T=$(echo $S | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $O/pmcnt_${T}_$C -o p -- python3 tools/pmc_gemm.py nt $S > $O/pmcnt.log 2>&1 || { echo "FAIL $S $C"; tail -3 $O/pmcnt.log; exit 1; }
    python3 tools/pmc_aggregate.py $O/pmcnt_${T}_$C $O/pmcnt_${T}_$C.json > /dev/null
    rm -rf $O/pmcnt_${T}_$C
  done
done
echo "pmc done"
# afterwards, in the repo: python3 tools/pmc_merge.py gpurun_out gpurun_out/r01_e_shape_breakdown.txt profiles/r01_e_pmc_gemm_nt.json
# and copy r01_e_bench_line.json, r01_e_event_breakdown.json, r01_e_shape_breakdown.txt, prof_r01e/stats_kernel_stats.csv into profiles/
