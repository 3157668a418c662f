cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  timeout -k 10 300 python3 tools/policy_time.py "" 2>&1 | grep -v amdgpu
  AZ_LIB=aozora_sdxl_training_amd/lib_exp_nt.so timeout -k 10 300 python3 tools/policy_time.py "" 2>&1 | grep -v amdgpu | sed 's/^/nt-stores  /'
done
