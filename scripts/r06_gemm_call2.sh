#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
{
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/microbench/mfma_f64_peak.hip && /tmp/mfma_peak
for v in 1 257 513; do
  echo "== variant $v cfg 0"
  TPA_GEMM_VARIANT=$v GEMM_CFG=0 DENSE=4096,4160,4224,8320 CHIS= timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "== variant 1025 cfg 1"
TPA_GEMM_VARIANT=1025 GEMM_CFG=1 DENSE=4096,4160,4224,8320 CHIS= timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids
} > $O/gemm_v2_call2.log 2>&1
cat $O/gemm_v2_call2.log
