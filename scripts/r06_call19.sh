#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/eigh570 -o eigh570 -- python $GRAFT_REPO_ROOT/scripts/eigh_direct_bench.py real 570 8 flat > $GRAFT_REPO_ROOT/$O/eigh570.log 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids $O/eigh570.log | tail -4
python scripts/kstats.py $O/eigh570/*kernel_stats.csv 2>/dev/null | head -30 || head -30 $O/eigh570/*kernel_stats.csv
