#!/bin/bash
# A/B of the split-K knob of the grouped GEMM (TPA_GEMM_SPLIT_K = "max parts,target tiles,min K per part") on the bench configurations.
# Usage (GPU box): bash scripts/split_k_ab.sh <config> <steps> <warmup> <knob> [<knob> ...]
cfg=$1; steps=$2; warm=$3; shift 3
mkdir -p gpurun_out
for k in "$@"; do
    TPA_GEMM_SPLIT_K=$k timeout 600 python bench.py --config $cfg --steps $steps --warmup $warm --no-extras --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/split_k_line.json
    python - "$cfg" "$k" <<'PY'
import json, sys
d = json.load(open('gpurun_out/split_k_line.json'))
g = d.get('roofline_gemm') or {}
print("%s split-K %s: %.4f %s; GEMM frac %.4f, %.4f ms/launch x %d, share %.3f; SVD %.3f ms/call; energy_err %s" % (
    sys.argv[1], sys.argv[2], d['value'], d['unit'], g.get('frac', 0), g.get('avg_launch_ms', 0), g.get('launches', 0),
    g.get('time_share_of_timed_region', 0), d['roofline'].get('avg_launch_ms', 0), d.get('energy_err')), flush=True)
PY
done
