#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06f
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 1800 python bench.py --gpus 1 --steps 10 --warmup 5 > $O/r06_bench_heis2048.json 2> $O/r06_bench_heis2048.err
tail -1 $O/r06_bench_heis2048.json > $O/r06_bench_heis2048_line.json
wc -c $O/r06_bench_heis2048_line.json
python -c "
import json
d=json.load(open('$O/r06_bench_heis2048_line.json'))
print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_gemm']['frac'])
print(json.dumps(d.get('other_configs')))
print(d.get('module_form'), d.get('force_dist'), d.get('lanczos_stats'), d.get('extras_s'))"
