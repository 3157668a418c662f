#!/bin/bash
# LayerNorm kernels: the round-3 tree (_ab_old) against this tree, rows per block 8 / 16 / 32
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rpb in 8 16 32; do
  echo "== old LN_RPB=$rpb"; (cd _ab_old && AZ_LN_RPB=$rpb timeout -k 10 120 python3 tools/norm_bench.py ln 40 2>&1 | grep "^ln")
  echo "== new LN_RPB=$rpb"; AZ_LN_RPB=$rpb timeout -k 10 120 python3 tools/norm_bench.py ln 40 2>&1 | grep "^ln"
done
