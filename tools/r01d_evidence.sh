#!/bin/bash
# Round-1 evidence run (on the GPU box): bench line, rocprofv3 kernel stats, two PMC passes (FETCH_SIZE, WRITE_SIZE).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py --steps 3 --warmup 1 --profile-out $O/r01_e_event_breakdown.json > $O/r01_e_bench_line.json 2> $O/r01_e_bench.err
echo "bench done"; cat $O/r01_e_bench_line.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01d -o stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_r01d.log 2>&1
echo "stats done"
rm -f $O/prof_r01d/*kernel_trace.csv $O/prof_r01d/*/*kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_r01d_$C -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_r01d_$C.log 2>&1
  python3 $R/tools/pmc_aggregate.py $O/pmc_r01d_$C $O/r01_e_pmc_$C.json
  rm -rf $O/pmc_r01d_$C
  echo "pmc $C done"
done
