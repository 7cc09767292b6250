#!/bin/bash
# Round 6, GPU call 1: baseline of the round-5 tree on today's box, the s_setprio experiment, dumps of the Jacobi inputs of
# steady-state warm / sketch calls (for scripts/warm_trace_emulate.py), cycle accounting of the solve (TPA_B32_TIMING build).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/benchA.log 2> $O/benchA.err
tail -c 3000 $O/benchA.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A', d['value'], d.get('roofline'), d.get('svd_stats'))"
TPA_B32_PRIO=1 TPA_SVD_DUMP_W=$O/W TPA_SVD_DUMP_MAX=5 TPA_SVD_DUMP_MIN_ROWS=500 TPA_SVD_DUMP_SKIP=450 TPA_SVD_DUMP_STRIDE=9 TPA_DUMP_THETA=$O/theta_chi2048_sat.npz \
  python bench.py --steps 4 --warmup 5 --no-extras --no-cpu-baseline > $O/benchB.log 2> $O/benchB.err
tail -c 3000 $O/benchB.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B(prio)', d['value'], d.get('roofline'), d.get('svd_stats'))"
ls -la $O/W $O
# cycle accounting of the solve on the dumped theta (cold call), timing build
if [ -f $O/theta_chi2048_sat.npz ]; then
  TPA_LIB_PATH=tenpy_amd/_lib_timing/libtenpy_amd.so REPS=1 CHECK=0 RHO=1e-2 timeout 300 python scripts/svd_file_bench.py $O/theta_chi2048_sat.npz > $O/timing_raw.log 2>&1
  grep "b32" $O/timing_raw.log | sort | uniq -c | sort -rn | head -60 > $O/timing_summary.txt
  grep "per iteration" $O/timing_raw.log | awk '{print $5, $6, $NF, $(NF-2)}' | sort | uniq -c | sort -rn | head -40 >> $O/timing_summary.txt
  tail -3 $O/timing_raw.log
  REPS=3 RHO=1e-2 python scripts/svd_file_bench.py $O/theta_chi2048_sat.npz | tail -3
  TPA_B32_PRIO=1 REPS=3 RHO=1e-2 python scripts/svd_file_bench.py $O/theta_chi2048_sat.npz | tail -3
fi
