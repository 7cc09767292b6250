#!/bin/bash
# final measurements of round 6 (tree with the Hermitian eigensolver): GPU test suite, the driver's bench command with its extras,
# kernel statistics of the TEBD eig route
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
O=gpurun_out/r06f
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 2400 python -m pytest tests -m gpu -q > $O/r06_gpu_tests.txt 2>&1
tail -4 $O/r06_gpu_tests.txt
timeout 1800 python bench.py --gpus 1 --steps 10 --warmup 5 > $O/r06_bench_heis2048.json 2> $O/r06_bench_heis2048.err
tail -1 $O/r06_bench_heis2048.json > $O/r06_bench_heis2048_line.json
cut -c1-5000 $O/r06_bench_heis2048_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_eig
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_eig -o b -- python $R/bench.py --config tebd1024 --qr --eig-svd --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/r06_bench_tebd1024_qr_eig_under_rocprof.json 2> /tmp/pt_eig.err < /dev/null
f=$(find /tmp/pt_eig -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r06_bench_tebd1024_qr_eig_kernel_stats.csv
head -12 $R/$O/r06_bench_tebd1024_qr_eig_kernel_stats.csv | cut -c1-200
