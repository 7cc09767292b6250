#!/bin/bash
# Round 6, GPU call 4: clean-up iterations (triangular panels), rule A/B with the dynamic schedule on; full parity sample.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_svd_warm.py tests/test_svd_rule.py tests/test_svd_configs_gpu.py tests/test_kernels_gpu.py tests/test_npc_completions.py -m gpu -q > $O/call4_tests.log 2>&1
tail -6 $O/call4_tests.log
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 4 --warmup 5 --no-extras --cpu-sample-bonds 1 > $O/bench_$name.log 2> $O/bench_$name.err
  tail -c 6000 $O/bench_$name.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], d.get('svd_stats'), {k: d.get(k) for k in ('energy_err','E','sv_max_rel_err','sv_max_rel_err_individual','svd_isometry_defect','mps_isometry_defect','sv_kept_rel_err_over_1e-10')})"
}
run min6 TPA_SVD_FLOOR_ON_MIN=1
run min5 TPA_SVD_FLOOR_ON_MIN=1 TPA_SVD_LOWDIN_ITERATIONS=5
run max4 TPA_SVD_FLOOR_ON_MIN=0
