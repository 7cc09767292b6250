#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
for alg in 0 4; do
echo "ALG=$alg"
ALG=$alg timeout 300 python scripts/eigh_direct_bench.py complex 1024 64 flat 2>&1 | grep -v amdgpu.ids | head -2
ALG=$alg timeout 300 python scripts/eigh_direct_bench.py real 1086 4 graded 2>&1 | grep -v amdgpu.ids | head -2
TPA_SVD_ALG0=$alg timeout 900 python bench.py --config tebd1024 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_tebd_alg$alg.log 2> $O/bench_tebd_alg$alg.err
tail -1 $O/bench_tebd_alg$alg.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('tebd svd route alg $alg', d['value'], json.dumps(d.get('roofline'))[:300], json.dumps(d.get('tebd_parity'))[:400], d.get('svd_stats'))"
done
