#!/bin/bash
# Round 6, GPU call 2: the stopping rule with the floor on the smaller row + ordered clean-up: SVD tests on the device, A/B of the bench.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_svd_warm.py tests/test_svd_rule.py tests/test_svd_configs_gpu.py tests/test_kernels_gpu.py tests/test_npc_completions.py -m gpu -x -q > $O/call2_tests.log 2>&1
tail -5 $O/call2_tests.log
for mode in 0 1; do
  TPA_SVD_FLOOR_ON_MIN=$mode python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_min$mode.log 2> $O/bench_min$mode.err
  tail -c 4000 $O/bench_min$mode.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('floor_on_min=$mode', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], d.get('svd_stats'), {k: d.get(k) for k in ('sv_max_rel_err','svd_isometry_defect','mps_isometry_defect','energy_err','E','sv_kept_rel_err_over_1e-10','matvec_max_rel_err')})"
done
