mkdir -p gpurun_out/r02s
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r02s/pytest_gpu.log; cat gpurun_out/r02s/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2 > gpurun_out/r02s/smoke.log; cat gpurun_out/r02s/smoke.log
timeout 900 python bench.py > gpurun_out/r02s/bench_heis2048.json 2> gpurun_out/r02s/bench_heis2048.err; tail -c 300 gpurun_out/r02s/bench_heis2048.json
for c in xxz512 hubbard1024 tebd1024; do
  timeout 900 python bench.py --config $c > gpurun_out/r02s/bench_$c.json 2> gpurun_out/r02s/bench_$c.err; tail -c 200 gpurun_out/r02s/bench_$c.json; echo
done
TPA_BENCH_PHASES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02s/bench_heis2048_phases.json 2> /dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02s/bench_under_rocprof.json 2> /dev/null
f=$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1); cp $f $R/gpurun_out/r02s/bench_kernel_stats.csv
CHECK=0 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o svd -- python $R/scripts/svd_file_bench.py > $R/gpurun_out/r02s/svd_file_under_rocprof.log 2>&1
f=$(find /tmp/prof_s -name '*kernel_stats.csv' | head -1); cp $f $R/gpurun_out/r02s/svd_call_kernel_stats.csv
