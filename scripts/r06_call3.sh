#!/bin/bash
# Round 6, GPU call 3: activity-driven rounds (dynamic schedule) of the Gram-only sweeps: SVD tests, A/B of the bench.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_svd_warm.py tests/test_svd_rule.py tests/test_svd_configs_gpu.py tests/test_kernels_gpu.py tests/test_npc_completions.py -m gpu -q > $O/call3_tests.log 2>&1
tail -8 $O/call3_tests.log
for alg in 16777216 0; do
  TPA_SVD_ALG0=$alg timeout 600 python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_dyn$alg.log 2> $O/bench_dyn$alg.err
  tail -c 5000 $O/bench_dyn$alg.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('alg0=$alg', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], d.get('svd_stats'), {k: d.get(k) for k in ('energy_err','E')})"
  tail -3 $O/bench_dyn$alg.err
done
