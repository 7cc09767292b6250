#!/bin/bash
# does a leg of the extras run slower after other legs in the same process?  hubbard1024 alone, after heis2048's sweeps, and after them with the caches cleared
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
python - <<'PY' 2>&1 | grep -v amdgpu | tail -8
import gc, sys, time
import bench
def leg(name):
    r = bench.run(["--config", name, "--steps", "2", "--warmup", "2", "--no-extras", "--no-cpu-baseline"], emit=False)
    return r["value"]
print("fresh hubbard", leg("hubbard1024"), flush=True)
print("fresh xxz512", leg("xxz512"), flush=True)
r = bench.run(["--steps", "2", "--warmup", "3", "--no-extras", "--no-cpu-baseline"], emit=False)
print("heis2048", r["value"], flush=True)
print("after heis: hubbard", leg("hubbard1024"), flush=True)
print("after heis: xxz512", leg("xxz512"), flush=True)
from tenpy_amd.linalg import np_conserved as npc
import torch
npc.clear_device_caches(); gc.collect(); torch.cuda.empty_cache()
print("after clear: hubbard", leg("hubbard1024"), flush=True)
print("after clear: xxz512", leg("xxz512"), flush=True)
PY
