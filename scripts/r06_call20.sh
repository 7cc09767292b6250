#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 300 python scripts/eigh_direct_bench.py complex 1024 64 flat 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py real 1086 4 graded 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py real 1086 2 graded14 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py real 570 8 flat 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_eig_svd.py tests/test_tebd_golden.py tests/test_kernels_gpu.py tests/test_npc_golden.py tests/test_npc_random.py tests/test_module_form_gpu.py -k "eig or tebd or mixer" -m gpu -q 2>&1 | tail -8
