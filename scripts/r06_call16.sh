#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python -m pytest tests/test_sharded.py -m gpu -q 2>&1 | tail -3
for i in 1 2; do
timeout 900 python bench.py --force-dist --steps 3 --warmup 4 --no-extras --no-cpu-baseline > $O/bench_fd$i.log 2> $O/bench_fd$i.err
grep '^{"metric"' $O/bench_fd$i.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('fd', d['value'], 'svd', d['roofline']['avg_launch_ms'], 'E', d['E'], d['energy_err'], d.get('lanczos_stats'))"
done
timeout 900 python bench.py --steps 3 --warmup 4 --no-extras --no-cpu-baseline > $O/bench_nofd.log 2> $O/bench_nofd.err
tail -1 $O/bench_nofd.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('plain', d['value'], 'svd', d['roofline']['avg_launch_ms'], 'E', d['E'], d['energy_err'])"
