#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 900 python bench.py --force-dist --steps 3 --warmup 4 --no-extras --no-cpu-baseline > $O/bench_fd.log 2> $O/bench_fd.err
tail -1 $O/bench_fd.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('fd', d['value'], 'svd', d['roofline']['avg_launch_ms'], d['roofline']['time_share_of_timed_region'], 'gemm', d['roofline_gemm']['frac'], d['roofline_gemm']['time_share_of_timed_region'], d['svd_stats'], d.get('lanczos_stats'))"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pfd; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfd -o b -- python $R/bench.py --force-dist --steps 2 --warmup 4 --no-cpu-baseline --no-extras > $R/$O/fd_under_rocprof.json 2> /tmp/ofd.txt < /dev/null
f=$(find /tmp/pfd -name "*kernel_trace.csv" | head -1)
SPAN=$(tail -c 4000 $R/$O/fd_under_rocprof.json | tail -1 | python -c "import sys,json; print(2*json.loads(sys.stdin.read())['value'])")
cd $R
[ -n "$f" ] && python scripts/gap_analysis.py "$f" $SPAN > $O/fd_gaps.txt 2>&1
[ -n "$f" ] && python scripts/trace_window.py "$f" $SPAN 30 > $O/fd_window.txt 2>&1
head -40 $O/fd_gaps.txt; head -36 $O/fd_window.txt
