#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_heff.py tests/test_dmrg_golden.py tests/test_split_k.py tests/test_percall_golden.py -m gpu -q -x > $O/gemm_call5_tests.log 2>&1
tail -4 $O/gemm_call5_tests.log
run() { echo "== $*"; env "$@" timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids; }
{
run DENSE=4096,8320 CHIS=2048,1024,512
run TPA_GEMM_VARIANT=1 DENSE= CHIS=1024,512
run TPA_GEMM_VARIANT=3 DENSE= CHIS=1024,512
} > $O/gemm_v2_call5.log 2>&1
cat $O/gemm_v2_call5.log | cut -c1-300
timeout 900 python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_gemm2.log 2> $O/bench_gemm2.err
tail -1 $O/bench_gemm2.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], 'svd ms', d['roofline']['avg_launch_ms'], 'gemm', d['roofline_gemm']['frac'], d['roofline_gemm']['time_share_of_timed_region'], d.get('svd_stats'), {k: d.get(k) for k in ('energy_err','E')})"
