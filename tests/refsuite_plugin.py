"""pytest plugin (``-p refsuite_plugin``) that lets the REFERENCE's own test files run, unedited, against the device
mirror: before any test module imports ``tenpy`` it registers the import hook of ``tenpy_amd.install`` (so that
``tenpy.linalg.np_conserved`` / ``.charges`` ARE ``tenpy_amd.linalg.np_conserved`` / ``.charges``) and -- when no GPU
is visible, i.e. in the CPU container -- the numpy emulation of the device entry points (``tests/mock_device.py``).
Used by ``tests/test_reference_suite.py``; test infrastructure only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


class _Setter:
    """The part of pytest's monkeypatch that mock_device.install uses, without undo (the mock stays for the session)."""
    @staticmethod
    def setattr(obj, name, value, raising=True):
        setattr(obj, name, value)


def _activate():
    import torch
    if not torch.cuda.is_available():
        import mock_device
        mock_device.install(_Setter)
    import tenpy_amd.install as ti
    ti.install(fused=bool(os.environ.get('TPA_REFSUITE_FUSED')))      # fused callers: test_reference_suite.py::test_reference_tebd_with_fused_callers
    import tenpy
    import tenpy_amd.linalg.np_conserved as mirror
    assert tenpy.linalg.np_conserved is mirror, "import hook not active"


_activate()
