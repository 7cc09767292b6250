#!/bin/bash
# round 6: the v2 GEMM loop (tpa_gemm.hip, gemm_chain2_kernel) against the old one: correctness, dense and matvec rates
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
for v in 1 257 513 1025; do
  echo "== variant $v"
  TPA_GEMM_VARIANT=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" 2>&1 | tail -3
  for cfg in 0 1; do
    echo "-- cfg $cfg"
    TPA_GEMM_VARIANT=$v GEMM_CFG=$cfg DENSE=4096 CHIS=512,2048 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  done
done > $O/gemm_v2_call1.log 2>&1
cat $O/gemm_v2_call1.log
