"""Which call sites upload / download per bond update in a steady-state sweep (stand-alone driver on the GPU):
python scripts/xfer_sites.py L chi n_warm"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
logging.disable(logging.WARNING)
import numpy as np
from tenpy_amd.linalg import _device as dev
cnt = collections.Counter()
byt = collections.Counter()


def wrap(name):
    f = getattr(dev, name)

    def g(*a, **k):
        st = traceback.extract_stack(limit=5)
        site = ' <- '.join('%s:%d' % (s.filename.split('/')[-1], s.lineno) for s in reversed(st[:-1]))
        cnt[(name, site)] += 1
        if name == 'to_device' and len(a) and hasattr(a[0], 'nbytes'):
            byt[(name, site)] += np.asarray(a[0]).nbytes
        return f(*a, **k)
    setattr(dev, name, g)


for n in ('to_device', 'to_device_packed', 'to_host', 'read_scalar', 'zeros', 'clone', 'take'):
    wrap(n)
from tenpy_amd.models.spin_chains import xxz_chain_mpo, spin_half_leg
from tenpy_amd.networks.mps import MPS
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
L, chi, nw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
H = xxz_chain_mpo(L, 1., 1., 0.)
chinfo, p = spin_half_leg('Sz')
psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
chi_list, c, s = {0: 64}, 64, 2
while c < chi:
    c = min(2 * c, chi)
    chi_list[s] = c
    s += 1
eng = TwoSiteDMRGEngine(psi, H, {'chi_list': chi_list, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-14}, 'lanczos_params': {'N_min': 8, 'N_max': 8}})
for s in range(len(chi_list) + 1 + nw):
    eng.sweep()
cnt.clear()
byt.clear()
eng.sweep()
nb = 2 * (L - 2)
tot = collections.Counter()
for (name, site), c in cnt.most_common(30):
    print('%6.2f per bond  %8.1f KB  %-10s %s' % (c / nb, byt[(name, site)] / max(c, 1) / 1e3, name, site))
for (name, site), c in cnt.items():
    tot[name] += c
print({k: round(v / nb, 1) for k, v in tot.items()})
