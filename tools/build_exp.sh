#!/bin/bash
# Experiment build of the library: bash tools/build_exp.sh <tag> [extra -D flags]  ->  aozora_sdxl_training_amd/lib_exp_<tag>.so
# (-DAZ_EXP_MINIMAL: only the default tile variants of the GEMM core are instantiated, so az_gemm.hip compiles in seconds;
#  the other objects are taken from the regular build).  For tools/gemm_ab, which loads several builds into one process.
set -e
cd "$(dirname "$0")/../aozora_sdxl_training_amd/csrc"
TAG=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -falign-loops=64 \
  -DAZ_EXP_MINIMAL "$@" -c az_gemm.hip -o /tmp/az_gemm_exp_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_exp_$TAG.so /tmp/az_gemm_exp_$TAG.o az_attn.o az_norm.o az_elem.o az_optim.o az_runtime.o az_tape.o
ls -la ../lib_exp_$TAG.so
