cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export REPS=3 CHECK=0
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $R/scripts/svd_file_bench.py > /tmp/o1.txt 2>&1 < /dev/null
grep "alg=" /tmp/o1.txt > $O/r04_svd_call_stdout.txt
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_svd_call_kernel_stats.csv
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $R/scripts/svd_file_bench.py > /tmp/o2.txt 2>&1 < /dev/null
f=$(find /tmp/p2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_svd_call_pmc_FETCH_SIZE.csv
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $R/scripts/svd_file_bench.py > /tmp/o3.txt 2>&1 < /dev/null
f=$(find /tmp/p3 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_svd_call_pmc_WRITE_SIZE.csv
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o b -- python bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-extras > $O/r04_bench_heis2048_under_rocprof.json 2> /tmp/o4.txt < /dev/null
f=$(find /tmp/p4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_bench_heis2048_kernel_stats.csv
ls -la $O | grep r04_ | head -20
