"""Build the C-ABI shared library ``libtenpy_amd.so`` (HIP kernels for gfx950 + C++ host planner).

In-tree build with explicit ``hipcc`` calls (no JIT cache): the resulting ``.so`` is git-ignored but
travels with the repo snapshot to the GPU box.  ``python -m tenpy_amd._build`` rebuilds.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.environ.get("TPA_BUILD_OUT") or os.path.join(HERE, "_lib")      # (TPA_BUILD_OUT + TPA_BUILD_FLAGS: a build variant beside the default, loaded with TPA_LIB_PATH)
LIB_PATH = os.path.join(OUT_DIR, "libtenpy_amd.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["tpa_gemm.hip", "tpa_vec.hip", "tpa_copy.hip", "tpa_svd.hip", "tpa_svd_theta.hip", "tpa_qr.hip", "tpa_util.hip",
           "tpa_plan.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("TPA_BUILD_FLAGS", "").split()


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    """Compile every source for gfx950 and link ``libtenpy_amd.so``.  Returns the library path."""
    os.makedirs(OUT_DIR, exist_ok=True)
    # every header / include file under csrc/ (a changed .inc must rebuild its .hip: round 4 lost a GPU run to a stale object)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    headers.append(os.path.join(HERE, "..", "include", "tenpy_amd.h"))
    objs = []
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if s.endswith(".hip"):
                cmd.insert(1, "--offload-arch=" + ARCH)
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB_PATH):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
