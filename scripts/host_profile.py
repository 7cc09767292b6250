"""Dev harness (GPU): cProfile of the HOST side of the timed chi = 2048 sweeps (driver protocol of bench.py): where the interpreter
spends its time between the launches, blocking read-backs included.  python scripts/host_profile.py [n_warm] [n_profiled]"""
import cProfile
import io
import os
import pstats
import sys
import types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = types.SimpleNamespace(L=100, chi=int(os.environ.get('CHI', 2048)), lanczos_N=8)
eng, _ = bench.build_dmrg(args, 1, 'heis2048')
eng.trunc_params['chi_max'] = args.chi
n_warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_prof = int(sys.argv[2]) if len(sys.argv) > 2 else 2
import time
import torch
for _ in range(n_warm):
    eng.sweep()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for _ in range(n_prof):
    eng.sweep()
torch.cuda.synchronize()
pr.disable()
print("profiled %d sweeps: %.3f s per sweep (under cProfile)" % (n_prof, (time.time() - t0) / n_prof))
for key, n in (('tottime', 70), ('cumulative', 70)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print("==== by", key)
    print(s.getvalue()[:14000])
