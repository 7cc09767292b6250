#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --config xxz512 --steps 4 --warmup 3 --no-cpu-baseline --no-extras > $O/x_$tag.log 2>/dev/null; tail -1 $O/x_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', d['value'])"; }
run default A=1
run omp1 OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1
run default2 A=1
run omp1b OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1
nproc; python -c "import torch; print(torch.get_num_threads())"
