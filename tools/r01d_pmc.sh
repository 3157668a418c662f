#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R && timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "transpose" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd $R && timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_r01d_$C -o p -- python3 tools/pmc_step.py 2 > $O/pmc_r01d_$C.log 2>&1 )
  python3 $R/tools/pmc_aggregate.py $O/pmc_r01d_$C $O/r01_d_pmc_$C.json
  rm -rf $O/pmc_r01d_$C
  echo "pmc $C done"
done
