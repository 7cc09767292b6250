#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1500 python -m pytest tests/test_svd_highprec.py tests/test_svd_warm.py tests/test_svd_rule.py tests/test_svd_configs_gpu.py tests/test_kernels_gpu.py tests/test_npc_completions.py tests/test_dmrg_golden.py tests/test_heff.py -m gpu -q > $O/call5_tests.log 2>&1
tail -6 $O/call5_tests.log
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$name.log 2> $O/bench_$name.err
  tail -c 6000 $O/bench_$name.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], d.get('svd_stats'), {k: d.get(k) for k in ('energy_err','E')})"
}
run c5 A=1
run c5_noround0 TPA_SVD_ALG0=33554432
python scripts/host_profile.py 5 2 > $O/host_profile2.txt 2>&1
grep -v poorly $O/host_profile2.txt | head -40 | cut -c1-150
