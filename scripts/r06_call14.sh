#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm 2>&1 | tail -3
DENSE=4096 CHIS=2048,1024,512 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-260
