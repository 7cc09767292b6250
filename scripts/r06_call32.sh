#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 900 python bench.py --config $cfg --steps 3 --warmup 5 --no-cpu-baseline --no-extras > $O/lc_$tag.log 2>/dev/null; tail -1 $O/lc_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['avg_launch_ms'], d.get('energy_err'), d.get('svd_stats',{}).get('max_block'), d.get('svd_stats',{}).get('sweeps_per_call'))"; }
run hub_on hubbard1024 TPA_SVD_LAYOUT_CACHE=1
run hub_off hubbard1024 TPA_SVD_LAYOUT_CACHE=0
run hub_on2 hubbard1024 TPA_SVD_LAYOUT_CACHE=1
run hub_off2 hubbard1024 TPA_SVD_LAYOUT_CACHE=0
