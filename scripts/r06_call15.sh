#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pfd; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfd -o b -- python $R/bench.py --force-dist --steps 2 --warmup 4 --no-cpu-baseline --no-extras > $R/$O/fd_under_rocprof.json 2> /tmp/ofd.txt < /dev/null
f=$(find /tmp/pfd -name "*kernel_trace.csv" | head -1)
SPAN=$(grep '^{"metric"' $R/$O/fd_under_rocprof.json | tail -1 | python -c "import sys,json; print(2*json.loads(sys.stdin.read())['value'])")
echo SPAN $SPAN
cd $R
python scripts/gap_analysis.py "$f" $SPAN > $O/fd_gaps.txt 2>&1
python scripts/trace_window.py "$f" $SPAN 30 > $O/fd_window.txt 2>&1
python scripts/trace_excerpt.py "$f" 1.3 400 > $O/fd_excerpt.txt 2>&1
head -30 $O/fd_gaps.txt; head -30 $O/fd_window.txt
