#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
for spec in graded diag; do
timeout 300 python scripts/eigh_bench.py 1086,871,871,450,450,148,148 $spec 2>&1 | grep -v amdgpu.ids | head -2
done
timeout 300 python scripts/eigh_bench.py 292,292,146,146,60 graded 2>&1 | grep -v amdgpu.ids | head -2
timeout 1800 python -m pytest tests/test_eig_svd.py tests/test_kernels_gpu.py tests/test_module_form_gpu.py tests/test_npc_golden.py tests/test_npc_random.py tests/test_reference_suite.py -k "eig or mixer or np_conserved or dmrg" -m gpu -q 2>&1 | tail -5
