#!/bin/bash
# PMC (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant GEMM class on the model's own shapes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
AZ_SHAPES=1 AZ_TOP=80 timeout -k 10 300 python3 tools/class_breakdown.py > $O/shapes_r01d.log 2>&1 || exit 1
cd /tmp; export TMPDIR=/tmp; cd $R
for S in "4096 10240 1280" "4096 1280 10240" "4096 1280 1280" "4096 1280 5120" "4096 5120 1280" "4096 3840 1280" "4096 1280 3840" \
         "16384 640 640" "16384 5120 640" "16384 640 5120" "16384 640 2560" "16384 2560 640" "16384 1920 640" "16384 640 1920"; do
  T=$(echo $S | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $O/pmcnt_${T}_$C -o p -- python3 tools/pmc_gemm.py nt $S > $O/pmcnt.log 2>&1 || { echo "FAIL $S $C"; tail -3 $O/pmcnt.log; exit 1; }
    python3 tools/pmc_aggregate.py $O/pmcnt_${T}_$C $O/pmcnt_${T}_$C.json > /dev/null
    rm -rf $O/pmcnt_${T}_$C
  done
  echo "done $S"
done
