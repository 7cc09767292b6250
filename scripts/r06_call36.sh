#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06f
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 2400 python -m pytest tests -m gpu -q > $O/r06_gpu_tests.txt 2>&1
tail -4 $O/r06_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
