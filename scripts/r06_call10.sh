#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1500 python -m pytest tests/test_svd_warm.py tests/test_svd_configs_gpu.py tests/test_dmrg_golden.py tests/test_midsize_golden.py tests/test_svd_highprec.py -m gpu -q -x > $O/call10_tests.log 2>&1
tail -15 $O/call10_tests.log
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$name.log 2> $O/bench_$name.err
  tail -1 $O/bench_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], 'gemm', d['roofline_gemm']['frac'], d['svd_stats'], {k: d.get(k) for k in ('energy_err','E')})"
}
runc() {
  name=$1; cfg=$2; shift; shift
  env "$@" timeout 900 python bench.py --config $cfg --steps 2 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_$name.log 2> $O/bench_$name.err
  tail -1 $O/bench_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], {k: d.get(k) for k in ('energy_err','E')})"
}
run n1 TPA_SVD_THETA_NATIVE=1
run n0 TPA_SVD_THETA_NATIVE=0
runc x_n1 xxz512 TPA_SVD_THETA_NATIVE=1
runc x_n0 xxz512 TPA_SVD_THETA_NATIVE=0
runc x_n1b xxz512 TPA_SVD_THETA_NATIVE=1
runc h_n1 hubbard1024 TPA_SVD_THETA_NATIVE=1
runc h_n0 hubbard1024 TPA_SVD_THETA_NATIVE=0
