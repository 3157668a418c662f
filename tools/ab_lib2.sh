#!/bin/bash
# Same-box A/B of the product library against another build: bash tools/ab_lib2.sh <other.so> [rounds] ["policy/options"]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for r in $(seq 1 ${2:-3}); do
  timeout -k 10 300 python3 tools/policy_time.py "$3" 2>&1 | grep -v amdgpu | sed 's/^/product  /'
  AZ_LIB=$1 timeout -k 10 300 python3 tools/policy_time.py "$3" 2>&1 | grep -v amdgpu | sed "s|^|$(basename $1)  |"
done
} > $O/ab_lib2.txt 2>&1
cat $O/ab_lib2.txt
