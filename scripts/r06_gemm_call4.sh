#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
run() { echo "== $*"; env "$@" DENSE= timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids; }
{
run TPA_GEMM_VARIANT=1 GEMM_CFG=1 CHIS=2048,1024,512 REPS=40
run TPA_GEMM_VARIANT=1025 GEMM_CFG=1 CHIS=2048,1024,512 REPS=40
run TPA_GEMM_VARIANT=513 GEMM_CFG=0 CHIS=2048,1024,512 REPS=40
run TPA_GEMM_VARIANT=257 GEMM_CFG=0 CHIS=2048,1024,512 REPS=40
run TPA_GEMM_VARIANT=1 CHIS=2048,1024,512 REPS=40
} > $O/gemm_v2_call4.log 2>&1
cat $O/gemm_v2_call4.log
