#!/bin/bash
# A/B of the memoised SVD layouts (TPA_SVD_LAYOUT_CACHE) on one box
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 4 --warmup 5 --no-cpu-baseline --no-extras > $O/lc_$tag.log 2>/dev/null; tail -1 $O/lc_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['avg_launch_ms'], d.get('energy_err'), [u['s'] for u in d.get('untimed_sweeps',[])][-7:])"; }
run h_on heis2048 TPA_SVD_LAYOUT_CACHE=1
run h_off heis2048 TPA_SVD_LAYOUT_CACHE=0
run h_on2 heis2048 TPA_SVD_LAYOUT_CACHE=1
run h_off2 heis2048 TPA_SVD_LAYOUT_CACHE=0
run x_on xxz512 TPA_SVD_LAYOUT_CACHE=1
run x_off xxz512 TPA_SVD_LAYOUT_CACHE=0
run x_on2 xxz512 TPA_SVD_LAYOUT_CACHE=1
run x_off2 xxz512 TPA_SVD_LAYOUT_CACHE=0
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_svd_warm.py tests/test_svd_configs_gpu.py -m gpu -q -x 2>&1 | tail -3
