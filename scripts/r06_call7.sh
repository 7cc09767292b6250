#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
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
run gc1 TPA_SWEEP_GC_PAUSE=1
run gc0 TPA_SWEEP_GC_PAUSE=0
runc x_gc1 xxz512 TPA_SWEEP_GC_PAUSE=1
runc x_gc0 xxz512 TPA_SWEEP_GC_PAUSE=0
runc h_gc1 hubbard1024 TPA_SWEEP_GC_PAUSE=1
runc h_gc0 hubbard1024 TPA_SWEEP_GC_PAUSE=0
