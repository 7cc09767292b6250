#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_eig_svd.py -k "eigh" -m gpu -q 2>&1 | tail -6
