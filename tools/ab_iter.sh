#!/bin/bash
# iteration-level A/B of two library builds (product vs AZ_LIB): per-micro-step timeline of tools/iter_timeline.py, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
  echo "== product library"; ROUNDS=2 timeout -k 10 300 python3 tools/iter_timeline.py 2>&1 | grep "^micro-steps\|update_region\|^{"
  echo "== $AB_LIB"; AZ_LIB=$AB_LIB ROUNDS=2 timeout -k 10 300 python3 tools/iter_timeline.py 2>&1 | grep "^micro-steps\|^{"
done
