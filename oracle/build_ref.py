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

Round 3 (VERDICT r2 "run unmodified TeNPy on the real MI355X"): ``pack()`` zips the reference's ``tenpy`` package, its
``tests`` and ``examples`` into ``oracle/_ref/tenpy_ref.zip`` -- one git-ignored build output next to the compiled helper,
which travels with the ``gpurun`` snapshot like the built ``.so`` files.  ``reference_root()`` returns ``/root/reference``
where it exists and otherwise unpacks the archive into ``oracle/_ref/unpacked`` (GPU box), so that the module-form tests
(``tests/test_module_form_gpu.py``, ``tests/test_reference_suite.py``) and ``bench.py``'s ``cpu_baseline`` leg can import the
reference there.  Still nothing of the reference is tracked by git, and the product never touches it.
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


ZIP = os.path.join(OUT, 'tenpy_ref.zip')
UNPACKED = os.path.join(OUT, 'unpacked')


def available():
    return os.path.exists(PYX)


def pack(force=False):
    """Archive the reference (package + tests + examples, sources only) into ``oracle/_ref/tenpy_ref.zip``."""
    import zipfile
    if not os.path.isdir(os.path.join(REF, 'tenpy')):
        raise RuntimeError("reference sources not found under " + REF)
    os.makedirs(OUT, exist_ok=True)
    newest = 0.
    files = []
    for top in ('tenpy', 'tests', 'examples'):
        for root, dirs, names in os.walk(os.path.join(REF, top)):
            dirs[:] = [d for d in dirs if d != '__pycache__']
            for n in names:
                if n.endswith(('.pyc', '.so')):
                    continue
                full = os.path.join(root, n)
                files.append(full)
                newest = max(newest, os.path.getmtime(full))
    if not force and os.path.exists(ZIP) and os.path.getmtime(ZIP) >= newest:
        return ZIP
    tmp = ZIP + '.tmp'
    with zipfile.ZipFile(tmp, 'w', zipfile.ZIP_DEFLATED) as z:
        for full in sorted(files):
            z.write(full, os.path.relpath(full, REF))
    os.replace(tmp, ZIP)
    return ZIP


def reference_root():
    """Directory that holds the reference's ``tenpy/``, ``tests/`` and ``examples/``: ``/root/reference`` if present, else the
    unpacked archive (GPU box), else None."""
    if os.path.isdir(os.path.join(REF, 'tenpy')):
        return REF
    if not os.path.exists(ZIP):
        return None
    stamp = os.path.join(UNPACKED, '.stamp')
    if not (os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(ZIP)):
        import shutil
        import zipfile
        shutil.rmtree(UNPACKED, ignore_errors=True)
        os.makedirs(UNPACKED, exist_ok=True)
        with zipfile.ZipFile(ZIP) as z:
            z.extractall(UNPACKED)
        with open(stamp, 'w') as f:
            f.write('unpacked from tenpy_ref.zip\n')
    return UNPACKED


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
    """Make ``import tenpy`` (from the reference tree, see ``reference_root``) pick up the compiled helper through its own
    hook (``tools/optimization.py:326``: ``from ..linalg import _npc_helper``).  Call before ``import tenpy``."""
    so = build(verbose=False) if available() else SO
    if not os.path.exists(so):
        raise RuntimeError("compiled reference helper not found: " + so)
    root = reference_root()
    if root is None:
        raise RuntimeError("no reference tree: neither %s nor %s" % (REF, ZIP))
    if root not in sys.path:
        sys.path.insert(0, root)
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder(so))


if __name__ == '__main__':
    print("built", build(force='--force' in sys.argv))
    print("packed", pack(force='--force' in sys.argv))
