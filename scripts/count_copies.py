"""Where do the rocclr copyBuffer launches of a sweep come from?  (round-4 idle-gap analysis: ~3000 per chi = 2048 sweep, 15 per bond,
each in front of a host-bound gap.)  Runs bench.py's sweep on the numpy emulation of the device and counts, per Python call site, the
torch operations that become a hipMemcpyAsync on the device: Tensor.copy_, clone, cpu, to, contiguous and the uploads of
_device.to_device.  Call counts per bond do not depend on chi."""
import collections, contextlib, io, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _pytest.monkeypatch import MonkeyPatch
import mock_device
mp = MonkeyPatch()
mock_device.install(mp)
import torch
mp.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
mp.setattr(torch.cuda, 'set_device', lambda *a, **k: None)
from tenpy_amd.linalg import _device as dev

COUNT = collections.Counter()
ON = [False]


def site():
    st = traceback.extract_stack(limit=8)[:-2]
    keep = [f for f in st if 'tenpy_amd' in f.filename or 'bench.py' in f.filename]
    return ' <- '.join('%s:%d(%s)' % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(keep[-3:]))


def wrap(obj, name, label):
    orig = getattr(obj, name)

    def w(*a, **k):
        if ON[0]:
            COUNT[(label, site())] += 1
        return orig(*a, **k)
    mp.setattr(obj, name, w)


for n in ('copy_', 'clone', 'cpu', 'to', 'contiguous', 'index_select', 'tolist', 'item'):
    wrap(torch.Tensor, n, 'Tensor.' + n)
wrap(dev, 'to_device', 'to_device')
wrap(dev, 'to_host', 'to_host')
import bench
orig_sweep = None
sys.argv = ['bench.py', '--L', '24', '--chi', '48', '--steps', '2', '--warmup', '3', '--no-cpu-baseline', '--no-extras']
from tenpy_amd.algorithms import dmrg
orig = dmrg.TwoSiteDMRGEngine.sweep if hasattr(dmrg, 'TwoSiteDMRGEngine') else None
ns = [0]


def sweep(self, *a, **k):
    ns[0] += 1
    ON[0] = ns[0] > 3
    return orig(self, *a, **k)


dmrg.TwoSiteDMRGEngine.sweep = sweep
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
ON[0] = False
nb = 2 * 2 * (24 - 2)
print("timed sweeps: 2, bond updates: %d" % nb)
for (label, s), c in COUNT.most_common(40):
    print("%6.2f per bond  %-16s %s" % (c / nb, label, s))
