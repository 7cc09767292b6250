#!/bin/bash
# direct two-sided Hermitian iteration of tpa_eigh_batch
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 300 python scripts/eigh_direct_bench.py complex 1024 2 flat 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py complex 1024 64 flat 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py real 1086 4 graded 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py real 570 8 flat 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/eigh_direct_bench.py real 200 8 graded 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_eig_svd.py tests/test_tebd_golden.py tests/test_kernels_gpu.py tests/test_npc_golden.py tests/test_npc_random.py -k "eig or tebd" -m gpu -q 2>&1 | tail -5
for cfg in "--qr --eig-svd"; do
n=$(echo $cfg | tr -d ' -')
timeout 900 python bench.py --config tebd1024 $cfg --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_tebd_$n.log 2> $O/bench_tebd_$n.err
tail -1 $O/bench_tebd_$n.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$n', d['value'], json.dumps(d.get('roofline'))[:400], json.dumps(d.get('tebd_parity'))[:600])"
done
