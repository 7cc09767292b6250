#!/bin/bash
# Kernel trace of the driver protocol (ramp + 5 warm-up + 2 timed sweeps); per-kernel busy time and idle gaps of the last SPAN seconds.
#   bash scripts/r06_profile_bench.sh <tag> <span seconds> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
tag=$1; span=$2; shift; shift
export TPA_NO_AUTOBUILD=1
cd $R
rm -rf /tmp/pt_$tag
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$tag -o b -- python bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_${tag}_bench.json 2> /tmp/pt_$tag.err < /dev/null
f=$(find /tmp/pt_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prof_${tag}_kernel_stats.csv
f=$(find /tmp/pt_$tag -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  python scripts/trace_window.py "$f" $span 40 > $O/prof_${tag}_window.txt 2>&1
  python scripts/gap_analysis.py "$f" $span > $O/prof_${tag}_gaps.txt 2>&1
  python scripts/trace_neighbors.py "$f" $span tri_lower 2 3 > $O/prof_${tag}_cleanup.txt 2>&1
  python scripts/trace_excerpt.py "$f" 1.2 1500 > $O/prof_${tag}_excerpt.txt 2>&1
fi
tail -c 3000 $O/prof_${tag}_bench.json | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['avg_launch_ms'])"
head -45 $O/prof_${tag}_window.txt
head -30 $O/prof_${tag}_gaps.txt
