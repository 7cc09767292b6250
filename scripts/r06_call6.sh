#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1500 python -m pytest tests/test_svd_highprec.py tests/test_svd_warm.py tests/test_svd_rule.py tests/test_svd_configs_gpu.py tests/test_npc_completions.py -m gpu -q > $O/call6_tests.log 2>&1
tail -4 $O/call6_tests.log
timeout 1500 python bench.py --steps 6 --warmup 5 > $O/bench_full6.log 2> $O/bench_full6.err
tail -1 $O/bench_full6.log | cut -c1-6000
