ALGS=0,2048 REPS=5 timeout 120 python scripts/svd_file_bench.py 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
R=$PWD
cd /tmp && export TMPDIR=/tmp
CHECK=0 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o svd -- python $R/scripts/svd_file_bench.py > /dev/null 2>&1
f=$(find /tmp/prof_s -name '*kernel_stats.csv' | head -1); head -3 $f | cut -c1-60,140-330
cd $R
TPA_BENCH_PHASES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/bench.json 2> /dev/null; python -c "
import json; d=json.loads(open('/tmp/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['phases_s'], d['energy_err'], d['svd_stats'])"
