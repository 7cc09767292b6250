#!/bin/bash
# final measurements of round 6: GPU test suite, the driver's bench command with its extras
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06f
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 2400 python -m pytest tests -m gpu -q > $O/r06_gpu_tests.txt 2>&1
tail -4 $O/r06_gpu_tests.txt
timeout 1800 python bench.py --gpus 1 --steps 10 --warmup 5 > $O/r06_bench_heis2048.json 2> $O/r06_bench_heis2048.err
tail -1 $O/r06_bench_heis2048.json > $O/r06_bench_heis2048_line.json
cut -c1-5000 $O/r06_bench_heis2048_line.json
