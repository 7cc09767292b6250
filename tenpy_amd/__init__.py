"""tenpy_amd -- MI355X (gfx950) native backend for TeNPy's block-sparse hot path.

See DESIGN.md.  ``tenpy_amd.linalg.np_conserved`` mirrors the API of ``tenpy.linalg.np_conserved``
with all block data resident in HBM and all floating-point work in hand-written HIP kernels behind the
C-ABI of ``include/tenpy_amd.h``.
"""
__version__ = "0.1.0"
