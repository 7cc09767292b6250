"""Build the reference's ONE native component -- ``tenpy/linalg/_npc_helper.pyx`` (+ ``_cblas_mkl.pxd``) -- from the
sources where they lie under ``/root/reference`` into ``oracle/_ref/`` (git-ignored; nothing is copied into the repo).

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  The compiled helper is what a TeNPy user runs on the CPU ("compiled without
HAVE_MKL": it links ``scipy.linalg.cython_blas``, _npc_helper.pyx:62-63, i.e. the OpenBLAS of the scipy wheel); it is the
`cpu_baseline` of kind "reference" (``scripts/cpu_reference_baseline.py``) and a second opinion for the oracle.  The
product never loads it.

Recipe (what ``setup.py:66-77`` of the reference does, minus its build system): cythonize the .pyx with the reference's
compiler directives (``language_level=3``, ``embedsignature``; compile-time env ``HAVE_MKL=0``) to C++ in
``oracle/_ref/``, then one ``g++ -O3 -shared`` call against the numpy / python headers.

Use: ``python oracle/build_ref.py`` (no-op when the .so is newer than the .pyx), then ``oracle.build_ref.load()``
registers the module as ``tenpy.linalg._npc_helper`` so that an ``import tenpy`` (from /root/reference) finds it through
its own hook ``tools/optimization.py:262 use_cython``.
"""
import importlib.machinery
import importlib.util
import os
import subprocess
import sys
import sysconfig

REF = os.environ.get('TENPY_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
PYX = os.path.join(REF, 'tenpy', 'linalg', '_npc_helper.pyx')
SO = os.path.join(OUT, '_npc_helper' + sysconfig.get_config_var('EXT_SUFFIX'))
CPP = os.path.join(OUT, '_npc_helper.cpp')


def available():
    return os.path.exists(PYX)


def build(force=False, verbose=True):
    if not available():
        raise RuntimeError("reference sources not found under " + REF)
    os.makedirs(OUT, exist_ok=True)
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(PYX):
        return SO
    import numpy as np
    from Cython.Compiler import Options
    from Cython.Compiler.Main import CompilationOptions, compile as cy_compile
    opts = CompilationOptions(Options.default_options, cplus=True, output_file=CPP,
                              include_path=[os.path.dirname(PYX)],
                              compiler_directives={'language_level': 3, 'embedsignature': True},
                              compile_time_env={'HAVE_MKL': 0})
    res = cy_compile(PYX, options=opts, full_module_name='tenpy.linalg._npc_helper')
    if res.num_errors:
        raise RuntimeError("cython failed on " + PYX)
    cmd = ['g++', '-O3', '-shared', '-fPIC', '-std=c++11', '-w', '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION',
           '-I' + np.get_include(), '-I' + sysconfig.get_paths()['include'], CPP, '-o', SO]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return SO


class _Finder:
    """meta-path finder: ``tenpy.linalg._npc_helper`` -> the .so under oracle/_ref (the reference tree is read-only, so
    the extension cannot sit next to its .pyx)."""
    name = 'tenpy.linalg._npc_helper'

    def __init__(self, so):
        self.so = so

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.name:
            return None
        loader = importlib.machinery.ExtensionFileLoader(fullname, self.so)
        return importlib.util.spec_from_file_location(fullname, self.so, loader=loader)


def load():
    """Make ``import tenpy`` (from /root/reference) pick up the compiled helper through its own hook
    (``tools/optimization.py:326``: ``from ..linalg import _npc_helper``).  Call before ``import tenpy``."""
    so = build(verbose=False)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder(so))


if __name__ == '__main__':
    print("built", build(force='--force' in sys.argv))
