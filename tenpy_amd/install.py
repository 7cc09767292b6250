"""Run an UNMODIFIED TeNPy on device-resident Arrays: ``import tenpy_amd.install as ti; ti.install(); import tenpy``.

``north_star``: "keeping the tenpy.linalg.np_conserved Array API and ChargeInfo/LegCharge bookkeeping so
algorithms/dmrg.py and algorithms/tebd.py run unchanged".  TeNPy's callers only ever reach block data through the
``np_conserved`` / ``charges`` API (nothing in algorithms/ or networks/mps.py touches ``_data``), so the whole
backend swap is an *import hook*: when the interpreter is asked for ``tenpy.linalg.np_conserved`` or
``tenpy.linalg.charges`` it is handed ``tenpy_amd.linalg.np_conserved`` / ``.charges`` instead -- the API mirror whose
``Array`` keeps its blocks in one HBM arena and whose workers are the HIP kernels behind ``include/tenpy_amd.h``.
Everything else (``tenpy.algorithms``, ``tenpy.networks``, ``tenpy.models``, ``tenpy.linalg.krylov_based``,
``truncation``, ``sparse`` ...) is the reference's own code, imported from wherever TeNPy is installed and run as is.

The reference ships its own replacement hook, ``tools/optimization.py:262 use_cython``, which looks functions up in
``tenpy.linalg._npc_helper``; it is evaluated inside the two modules replaced here, so with the mirror in place it is
never consulted (``have_cython_functions`` is set to ``False``, the state the reference itself reaches with
``TENPY_NO_CYTHON=1``, optimization.py:322).  The fine-grained form of the boundary -- the 16 ``use_cython`` names for
a TeNPy that keeps its own ``np_conserved`` -- is ``tenpy_amd/_npc_helper.py``.

``install(fused=True)`` additionally rebinds, after ``import tenpy``, the callers for which the device has a
fused form: ``LanczosGroundState`` (one fused recurrence kernel per step instead of four BLAS-1 calls), ``TwoSiteH``
(cached plans; factored matvec LP . theta . W0 W1 . RP for ``combine=False``) and the bond hint of the warm-started block
SVD; see :func:`use_fused_callers`.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

__all__ = ['install', 'uninstall', 'installed', 'use_fused_callers']

_SUBST = {
    'tenpy.linalg.np_conserved': 'tenpy_amd.linalg.np_conserved',
    'tenpy.linalg.charges': 'tenpy_amd.linalg.charges',
}


class _MirrorLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        mod = importlib.import_module(self.target)
        self._orig_spec = getattr(mod, '__spec__', None)
        return mod

    def exec_module(self, module):
        # the import system has overwritten __spec__ / left __name__ alone; keep the mirror's own identity
        module.__spec__ = self._orig_spec
        # the reference evaluates @use_cython while importing the two modules replaced here and asserts afterwards
        # that somebody did (linalg/__init__.py:74); record "no compiled helper", like TENPY_NO_CYTHON=1
        from tenpy.tools import optimization
        if optimization.have_cython_functions is None:
            optimization.have_cython_functions = False


class _MirrorFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        tgt = _SUBST.get(fullname)
        if tgt is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, _MirrorLoader(tgt))


_finder = None


def installed():
    return _finder is not None


def install(fused=False):
    """Register the import hook.  Must run before the first ``import tenpy`` (raises otherwise: a TeNPy that has
    already bound its own ``np_conserved`` in dozens of module namespaces cannot be re-pointed reliably)."""
    global _finder
    if _finder is None:
        for name in _SUBST:
            if name in sys.modules and sys.modules[name].__name__ != _SUBST[name]:
                raise RuntimeError("tenpy was imported before tenpy_amd.install.install(); call install() first")
        _finder = _MirrorFinder()
        sys.meta_path.insert(0, _finder)
    if fused:
        use_fused_callers()


def uninstall():
    """Remove the hook and forget every ``tenpy`` module, so that a later ``import tenpy`` is the plain CPU TeNPy."""
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
    for name in [n for n in sys.modules if n == 'tenpy' or n.startswith('tenpy.')]:
        del sys.modules[name]


def use_fused_callers():
    """Optional, after the hook: rebind, in the reference's modules, the callers for which the device has a fused form.
    The engines themselves (``algorithms/dmrg.py``, ``mps_common.py`` sweeps, mixers, ``tebd.py``) remain the reference's code.

    * ``LanczosGroundState`` (``linalg/krylov_based.py:645``; constructed at ``dmrg.py:740/742``) -> the fused device recurrence
      of ``tenpy_amd/linalg/krylov_based.py`` (same options, same results, one kernel per Krylov step);
    * ``TwoSiteH`` (``algorithms/mps_common.py:1245``; ``TwoSiteDMRGEngine.EffectiveH``, ``dmrg.py:865``) -> the device form of
      ``tenpy_amd/algorithms/module_form.py`` (cached plans, fused ``LHeff`` build; ``combine=False``: factored matvec), which
      hands bonds it does not cover back to the reference's class;
    * ``TwoSiteDMRGEngine.mixed_svd`` (``dmrg.py:876``) is wrapped to pass the bond index to the block SVD (warm start).
    * ``TEBDEngine.evolve_step`` (``tebd.py:374``; inherited by ``QRBasedTEBDEngine``) -> the bonds of a half-step decomposed in one
      batched device call (``module_form.batched_tebd_evolve_step``; the per-bond statements stay the reference's).
    """
    import tenpy.algorithms.dmrg as ref_dmrg
    import tenpy.algorithms.mps_common as ref_mc
    import tenpy.linalg.krylov_based as ref_kb
    from .algorithms import module_form
    from .linalg import krylov_based as kb
    ref_kb.LanczosGroundState = kb.LanczosGroundState
    ref_dmrg.LanczosGroundState = kb.LanczosGroundState
    if not hasattr(ref_mc.TwoSiteH, '_reference_class'):
        dev_cls = module_form.device_two_site_h(ref_mc.TwoSiteH)
        ref_mc.TwoSiteH = dev_cls
        ref_dmrg.TwoSiteH = dev_cls
        ref_dmrg.TwoSiteDMRGEngine.EffectiveH = dev_cls
    if not getattr(ref_dmrg.TwoSiteDMRGEngine.mixed_svd, '_tpa_wrapped', False):
        ref_dmrg.TwoSiteDMRGEngine.mixed_svd = module_form.hinted_mixed_svd(ref_dmrg.TwoSiteDMRGEngine.mixed_svd)
    import tenpy.algorithms.tebd as ref_tebd
    if not getattr(ref_tebd.TEBDEngine.evolve_step, '_tpa_wrapped', False):
        ref_tebd.TEBDEngine.evolve_step = module_form.batched_tebd_evolve_step(ref_tebd)
