#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests_call23.txt
cat $O/gpu_tests_call23.txt
