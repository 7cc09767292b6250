#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1500 python -m pytest tests/test_svd_warm.py tests/test_svd_configs_gpu.py tests/test_dmrg_golden.py tests/test_midsize_golden.py tests/test_split_k.py tests/test_heff.py -m gpu -q > $O/call9_tests.log 2>&1
tail -3 $O/call9_tests.log
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$name.log 2> $O/bench_$name.err
  tail -1 $O/bench_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], 'gemm', d['roofline_gemm']['frac'], {k: d.get(k) for k in ('energy_err','E')})"
}
runc() {
  name=$1; cfg=$2; shift; shift
  env "$@" timeout 900 python bench.py --config $cfg --steps 2 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_$name.log 2> $O/bench_$name.err
  tail -1 $O/bench_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], {k: d.get(k) for k in ('energy_err','E')})"
}
run a A=1
run b A=2
runc x_a xxz512 A=1
runc x_b xxz512 A=2
runc h_a hubbard1024 A=1
runc h_b hubbard1024 A=2
