#!/bin/bash
# final tree of round 6: GPU test suite + the driver's bench command
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06f
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 2400 python -m pytest tests -m gpu -q > $O/r06_gpu_tests.txt 2>&1
tail -4 $O/r06_gpu_tests.txt
timeout 1800 python bench.py --gpus 1 --steps 10 --warmup 5 > $O/r06_bench_heis2048.json 2> $O/r06_bench_heis2048.err
tail -1 $O/r06_bench_heis2048.json > $O/r06_bench_heis2048_line.json
wc -c $O/r06_bench_heis2048_line.json
python -c "
import json
d=json.load(open('$O/r06_bench_heis2048_line.json'))
print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_gemm']['frac'], d['energy_err'])
print(json.dumps(d.get('other_configs')))
print(d.get('module_form'), d.get('force_dist'), d.get('lanczos_stats'), d.get('lanczos_adaptive'), d.get('first_sweeps_at_target_chi_s'), d.get('extras_s'))"
