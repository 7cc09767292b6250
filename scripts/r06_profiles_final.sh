# Round-6 profile passes (run on the MI355X box through gpurun; outputs land in gpurun_out/r06p/, the summaries are copied to profiles/).
#   bash scripts/r06_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O $R/scripts/data
export TPA_NO_AUTOBUILD=1
cd $R
# (1) kernel statistics + trace of the driver protocol (ramp + 5 warm-up + 2 timed sweeps): per-kernel busy time and idle gaps of the timed sweeps
rm -rf /tmp/p1; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-extras > $O/r06_bench_heis2048_under_rocprof.json 2> /tmp/o1.txt < /dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_bench_heis2048_kernel_stats.csv
f=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1)
SPAN=$(tail -c 4000 $O/r06_bench_heis2048_under_rocprof.json | tail -1 | python -c "import sys,json; print(2*json.loads(sys.stdin.read())['value'])")
if [ -n "$f" ]; then
  python scripts/gap_analysis.py "$f" $SPAN > $O/r06_idle_gap_analysis.txt 2>&1
  python scripts/trace_window.py "$f" $SPAN 45 > $O/r06_bench_heis2048_timed_sweeps_by_kernel.txt 2>&1
  python scripts/trace_excerpt.py "$f" 1.3 700 > $O/r06_bench_trace_excerpt.txt 2>&1
fi
# (2) the saturated theta of the centre bond, then the COLD call on it: kernel statistics, FETCH_SIZE / WRITE_SIZE in separate passes
TPA_DUMP_THETA=$R/scripts/data/theta_chi2048_sat.npz timeout 600 python bench.py --steps 1 --warmup 5 --no-extras --cpu-sample-bonds 1 > $O/r06_bench_dump_run.json 2> /tmp/o2.txt < /dev/null
export RHO=-1e-2 REPS=3 CHECK=0
cd /tmp
# (3) the other DMRG configurations under the tracer: busy share and gaps
for cfg in xxz512 hubbard1024; do
  rm -rf /tmp/p_$cfg; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$cfg -o h -- python bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/r06_bench_${cfg}_under_rocprof.json 2> /tmp/o_$cfg.txt < /dev/null
  f=$(find /tmp/p_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_bench_${cfg}_kernel_stats.csv
  f=$(find /tmp/p_$cfg -name "*kernel_trace.csv" | head -1)
  SPAN=$(tail -c 4000 $O/r06_bench_${cfg}_under_rocprof.json | tail -1 | python -c "import sys,json; print(2*json.loads(sys.stdin.read())['value'])")
  [ -n "$f" ] && python scripts/gap_analysis.py "$f" $SPAN > $O/r06_idle_gap_analysis_$cfg.txt 2>&1
done
