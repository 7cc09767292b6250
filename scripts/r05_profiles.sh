# Round-5 profile passes (run on the MI355X box through gpurun; outputs land in gpurun_out/, the summaries are copied to profiles/).
#   bash scripts/r05_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export CHECK=0
# (1) kernel statistics + kernel trace of the driver protocol (short: ramp + 3 warm-up + 2 timed sweeps), idle-gap analysis of the last sweep
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o b -- python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-extras > $O/r05_bench_heis2048_under_rocprof.json 2> /tmp/o4.txt < /dev/null
f=$(find /tmp/p4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_bench_heis2048_kernel_stats.csv
f=$(find /tmp/p4 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/gap_analysis.py "$f" 2.7 > $O/r05_idle_gap_analysis.txt 2>&1
# (2) the sketch call (stale basis) and the cold call on the saturated chi = 2048 theta: kernel statistics, then FETCH_SIZE / WRITE_SIZE in separate passes
cd /tmp
REPS=3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $R/scripts/svd_sketch_bench.py > $O/r05_sketch_call_stdout.txt 2>&1 < /dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_sketch_call_kernel_stats.csv
REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $R/scripts/svd_sketch_bench.py > /tmp/o2.txt 2>&1 < /dev/null
f=$(find /tmp/p2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_sketch_call_pmc_FETCH_SIZE.csv
REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $R/scripts/svd_sketch_bench.py > /tmp/o3.txt 2>&1 < /dev/null
f=$(find /tmp/p3 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_sketch_call_pmc_WRITE_SIZE.csv
# (2b) the COLD call alone (scripts/svd_file_bench.py, the file behind bench.py's roofline.traffic; engine floor 1e-4 as in the sweeps)
export RHO=1e-4 REPS=3
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q1 -o s -- python $R/scripts/svd_file_bench.py > $O/r05_svd_call_stdout.txt 2>&1 < /dev/null
f=$(find /tmp/q1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_svd_call_kernel_stats.csv
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q2 -o f -- python $R/scripts/svd_file_bench.py > /tmp/o6.txt 2>&1 < /dev/null
f=$(find /tmp/q2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_svd_call_pmc_FETCH_SIZE.csv
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q3 -o w -- python $R/scripts/svd_file_bench.py > /tmp/o7.txt 2>&1 < /dev/null
f=$(find /tmp/q3 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_svd_call_pmc_WRITE_SIZE.csv
# (3) the Hubbard ladder (many small blocks): kernel statistics of 1 warm-up + 2 timed sweeps
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o h -- python bench.py --config hubbard1024 --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/r05_bench_hubbard1024_under_rocprof.json 2> /tmp/o5.txt < /dev/null
f=$(find /tmp/p5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_bench_hubbard1024_kernel_stats.csv
f=$(find /tmp/p5 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/gap_analysis.py "$f" 1.7 > $O/r05_idle_gap_analysis_hubbard1024.txt 2>&1
ls -la $O | grep r05_ | head -30
