#!/bin/bash
# identity-row skip of the sweep-end product of the block SVD: tests, then A/B on one box (TPA_SVD_APPLY_SKIP)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_svd_warm.py tests/test_svd_configs_gpu.py tests/test_svd_highprec.py tests/test_eig_svd.py -m gpu -q -x 2>&1 | tail -4
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 4 --warmup 5 --no-cpu-baseline --no-extras > $O/sk_$tag.log 2>/dev/null; tail -1 $O/sk_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['avg_launch_ms'], d.get('energy_err'), d.get('sv_max_rel_err'), d.get('svd_isometry_defect'), d.get('svd_stats',{}).get('sweeps_per_call'))"; }
run h_on heis2048 TPA_SVD_APPLY_SKIP=1
run h_off heis2048 TPA_SVD_APPLY_SKIP=0
run h_on2 heis2048 TPA_SVD_APPLY_SKIP=1
run h_off2 heis2048 TPA_SVD_APPLY_SKIP=0
run hub_on hubbard1024 TPA_SVD_APPLY_SKIP=1
run hub_off hubbard1024 TPA_SVD_APPLY_SKIP=0
run x_on xxz512 TPA_SVD_APPLY_SKIP=1
run x_off xxz512 TPA_SVD_APPLY_SKIP=0
