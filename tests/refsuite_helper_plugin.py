"""pytest plugin (``-p refsuite_helper_plugin``): the reference keeps its OWN ``np_conserved`` / ``charges`` and gets
``tenpy_amd/_npc_helper.py`` as its native helper module through its own hook (``tools/optimization.py:262 use_cython``).
In the CPU container the device entry points are the numpy emulation.  Test infrastructure only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


class _Setter:
    @staticmethod
    def setattr(obj, name, value, raising=True):
        setattr(obj, name, value)


def _activate():
    import torch
    if not torch.cuda.is_available():
        import mock_device
        mock_device.install(_Setter)
    import tenpy_amd._npc_helper as helper
    helper.register()
    import tenpy
    from tenpy.tools import optimization
    assert optimization.have_cython_functions is True
    assert tenpy.linalg.np_conserved._tensordot_worker is helper._tensordot_worker
    assert tenpy.linalg.np_conserved.Array.itranspose is helper.Array_itranspose
    assert tenpy.linalg.charges.LegPipe._init_from_legs is helper.LegPipe__init_from_legs


_activate()
