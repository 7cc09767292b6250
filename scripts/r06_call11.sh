#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1500 python -m pytest tests/test_svd_warm.py tests/test_svd_configs_gpu.py tests/test_svd_highprec.py tests/test_dmrg_golden.py tests/test_midsize_golden.py tests/test_kernels_gpu.py -m gpu -q > $O/call11_tests.log 2>&1
tail -5 $O/call11_tests.log
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 4 --warmup 5 --no-extras --cpu-sample-bonds 1 > $O/bench_$name.log 2> $O/bench_$name.err
  tail -1 $O/bench_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'svd ms', d['roofline']['avg_launch_ms'], 'gemm', d['roofline_gemm']['frac'], {k: d.get(k) for k in ('energy_err','E','sv_max_rel_err','svd_isometry_defect','mps_isometry_defect','sv_max_rel_err_vs_highprec','matvec_max_rel_err')})"
}
runc() {
  name=$1; cfg=$2; shift; shift
  env "$@" timeout 900 python bench.py --config $cfg --steps 2 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_$name.log 2> $O/bench_$name.err
  tail -1 $O/bench_$name.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', d['value'], {k: d.get(k) for k in ('energy_err','E')})"
}
run g1 TPA_SVD_CLEAN_GRADED=1
run g0 TPA_SVD_CLEAN_GRADED=0
runc x_g1 xxz512 TPA_SVD_CLEAN_GRADED=1
runc x_g0 xxz512 TPA_SVD_CLEAN_GRADED=0
runc h_g1 hubbard1024 TPA_SVD_CLEAN_GRADED=1
runc h_g0 hubbard1024 TPA_SVD_CLEAN_GRADED=0
