#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06f
mkdir -p $O
export TPA_NO_AUTOBUILD=1
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -v amdgpu.ids | tail -5
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -1 $O/bench_default.json | cut -c1-700
