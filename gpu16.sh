mkdir -p gpurun_out/r02p
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_svd_rule.py tests/test_fullsize_gpu.py tests/test_npc_completions.py tests/test_percall_golden.py tests/test_midsize_golden.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02p/pytest_svd.log
cat gpurun_out/r02p/pytest_svd.log
ALGS=0,2048,0 REPS=5 timeout 300 python scripts/svd_file_bench.py > gpurun_out/r02p/svd_file.log 2>&1; cat gpurun_out/r02p/svd_file.log
TPA_BENCH_PHASES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02p/bench.json 2> gpurun_out/r02p/bench.err; tail -c 700 gpurun_out/r02p/bench.json
