"""Thin device helpers: arenas in HBM via torch (memory + streams only) and calls into the C-ABI.

torch is *plumbing* here (allocation, H2D/D2H copies, the current HIP stream); every arithmetic or
data-movement kernel on block data is one of the hand-written HIP kernels behind ``include/tenpy_amd.h``.
"""
import ctypes
import os

import numpy as np

from .. import _lib

_torch = None
_scratch = {}


def torch():
    global _torch
    if _torch is None:
        import torch as _t
        _torch = _t
    return _torch


def lib():
    """The loaded C-ABI library; raises if no GPU is visible (no CPU fallback)."""
    _lib.require_gpu()
    return _lib.load()


_raw_stream = None


def stream():
    """Raw handle of torch's CURRENT stream on the current device (so `torch.cuda.stream(...)` contexts are honoured).  Goes through
    torch's C entry point: `torch.cuda.current_stream().cuda_stream` builds a Python Stream object per call (9 us; ~100 calls per bond
    update = 0.19 s per sweep at any chi, measured with cProfile in round 2)."""
    global _raw_stream
    if _raw_stream is None:
        t = torch()
        fast = getattr(t._C, '_cuda_getCurrentRawStream', None)
        getdev = getattr(t._C, '_cuda_getDevice', None) or t.cuda.current_device
        _raw_stream = (lambda: fast(getdev())) if fast is not None else (lambda: t.cuda.current_stream().cuda_stream)
    return _raw_stream()


def code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return _lib.F64
    if dtype == np.complex128:
        return _lib.C128
    raise ValueError("tenpy_amd computes in float64 / complex128 only, got " + str(dtype))


def tdtype(dtype):
    t = torch()
    return t.float64 if np.dtype(dtype) == np.float64 else t.complex128


def empty(n, dtype):
    _lib.require_gpu()
    return torch().empty(int(n), dtype=tdtype(dtype), device='cuda')


def zeros(n, dtype):
    arena = empty(n, dtype)
    if n > 0:
        _lib.check(lib().tpa_fill_zero(arena.data_ptr(), int(n) * arena.element_size(), stream()), "fill_zero")
    return arena


_pool = {}
import threading as _threading  # noqa: E402
_thread_ident = _threading.get_ident
_main_thread = [_threading.main_thread().ident]


def scratch(name, n, dtype):
    """Persistent, grow-only scratch arena (1.5x over-allocation): a typed view of ``n`` elements.  For temporaries whose size
    changes from call to call (the work area of the block SVD, the intermediates of the warm start): asking torch's caching
    allocator for a different size every call splits its cached blocks and ends in a fresh hipMalloc per call (measured: the
    same ``tpa_svd_batch`` took 20.7 instead of 16.4 ms when 300 MB of temporaries had been allocated and freed before it).
    The contents are only valid until the next request under the same name; all users are ordered on one stream."""
    ident = _thread_ident()
    if ident != _main_thread[0]:
        name = (name, ident)          # a second host thread (DMRGThreadPlusHC) gets pools of its own
    n = int(n)
    ent = _pool.get(name)
    if ent is not None:               # the same (size, type) again -- every bond of a sweep asks ~24 times: the view made last time
        got = ent[2].get((n, dtype))
        if got is not None:
            return got
    dt = np.dtype(dtype)
    need = max(n, 1) * dt.itemsize
    if ent is None or ent[0].numel() * 8 < need:
        buf = empty(((int(need * 1.5) + 7) // 8 + 32) // 2 * 2, np.float64)      # `empty`: the (test-patchable) allocator; even: complex views
        ent = _pool[name] = (buf, {}, {})
    buf, views, sized = ent
    v = views.get(dt.char)
    if v is None:
        import torch as real_torch
        v = views[dt.char] = buf.view({'float64': real_torch.float64, 'complex128': real_torch.complex128, 'uint8': real_torch.uint8}[dt.name])
    out = v[:max(n, 1)] if n > 0 else v[:0]
    if len(sized) >= 64:
        sized.clear()
    sized[(n, dtype)] = out
    return out


PIN_RING_BYTES = 64 << 20
PIN_MAX_BYTES = 1 << 20
_pin_ring = None
_pin_lock = None


def to_device(arr):
    """numpy array (any int/float dtype) -> device tensor of the same dtype.

    Small arrays (tables, scale vectors, singular values: <= 1 MB) go through a persistent PINNED ring buffer and a non-blocking
    copy on the current stream: ``tensor.to('cuda')`` from pageable memory blocks the host for ~40 us per call, and the per-call
    tables of the warm start / Loewdin clean-up alone are 10 - 20 uploads per ``npc.svd`` (round-3 idle-gap analysis: the GPU sat
    idle behind them).  The source is copied into the ring before this returns, so the caller may reuse it at once; the ring wraps
    after a device synchronisation (every few hundred uploads).  Larger arrays: plain (blocking) copy."""
    global _pin_ring, _pin_lock
    _lib.require_gpu()
    t = torch()
    arr = np.ascontiguousarray(arr)
    nb = arr.nbytes
    if nb == 0 or nb > PIN_MAX_BYTES or arr.dtype.kind not in 'iufcb' or arr.dtype.itemsize > 16:
        return t.from_numpy(arr).to('cuda')
    tdt = t.from_numpy(np.empty(0, dtype=arr.dtype)).dtype
    if _pin_lock is None:
        import threading
        _pin_lock = threading.Lock()
    with _pin_lock:
        if _pin_ring is None:
            buf = t.empty(PIN_RING_BYTES, dtype=t.uint8).pin_memory()
            _pin_ring = [buf, buf.numpy(), 0]
        buf, view, pos = _pin_ring
        pos = (pos + 255) // 256 * 256
        if pos + nb > PIN_RING_BYTES:
            t.cuda.synchronize()          # every copy that was queued out of the ring has been executed
            pos = 0
        view[pos:pos + nb] = arr.reshape(-1).view(np.uint8)
        _pin_ring[2] = pos + nb
        dst = t.empty(nb, dtype=t.uint8, device='cuda')
        dst.copy_(buf[pos:pos + nb], non_blocking=True)
    return dst.view(tdt).reshape(arr.shape)


def to_device_packed(*arrays):
    """``tuple(to_device(a) for a in arrays)`` as ONE upload: the arrays are laid out back to back (256-byte aligned) in one staging
    buffer and the results are typed views of the one device buffer.  For the tables of a plan (tasks, links and tiles of a grouped
    GEMM ...): every upload is its own ``hipMemcpyAsync`` with ~40 us of host time around it (round-6 idle-gap analysis:
    ``copyBuffer -> copyBuffer`` gaps)."""
    import torch as real_torch
    arrs = [np.ascontiguousarray(a) for a in arrays]
    offs, pos = [], 0
    for a in arrs:
        pos = (pos + 255) // 256 * 256
        offs.append(pos)
        pos += a.nbytes
    stage = np.zeros(max(pos, 1), dtype=np.uint8)
    for a, o in zip(arrs, offs):
        if a.nbytes:
            stage[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    d = to_device(stage)
    out = []
    for a, o in zip(arrs, offs):
        tdt = real_torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        v = d[o:o + a.nbytes].view(tdt).reshape(a.shape)
        v._tpa_owner = d          # (the views share the owner's storage; this also keeps its Python object alive with them)
        out.append(v)
    return tuple(out)


_table_cache = None
TABLE_CACHE_MAX_ENTRIES = 32768
TABLE_CACHE_TOTAL_BYTES = 1 << 30     # of device memory for all cached tables together
_table_bytes = 0
TABLE_CACHE_MAX_BYTES = 1 << 18       # per table; larger ones are uploaded every time (hashing them would cost more than the copy)


def table(arr):
    """Device copy of a small READ-ONLY host table (job / index tables of the copy, scale, gather and combination kernels), cached
    by content.  A saturated DMRG / TEBD run issues the same few thousand tables again and again (one per reshaping operation and
    bond); each upload from pageable memory is a blocking ~40 us host call in front of a ~5 us kernel, and the round-3 idle-gap
    analysis (scripts/gap_analysis.py) attributes 0.3 of the 4.0 s of a chi = 2048 sweep to the GPU waiting for exactly these."""
    global _table_cache
    arr = np.ascontiguousarray(arr)
    if arr.nbytes > TABLE_CACHE_MAX_BYTES:
        return to_device(arr)
    if _table_cache is None:
        from collections import OrderedDict
        _table_cache = OrderedDict()
    global _table_bytes
    key = (arr.dtype.str, arr.shape, arr.tobytes())
    t = _table_cache.get(key)
    if t is None:
        t = _table_cache[key] = to_device(arr)
        _table_bytes += arr.nbytes
        while len(_table_cache) > TABLE_CACHE_MAX_ENTRIES or _table_bytes > TABLE_CACHE_TOTAL_BYTES:
            _, old = _table_cache.popitem(last=False)
            _table_bytes -= old.numel() * old.element_size()
    else:
        try:
            _table_cache.move_to_end(key)
        except KeyError:      # evicted by another thread in between (the reference's `+ h.c.` worker contracts concurrently)
            pass
    return t


def clone(tensor):
    return tensor.clone()


def take(tensor, index_array):
    """Gather a few scalars (e.g. a diagonal) from an arena; index_array is a host int64 array."""
    return tensor[to_device(np.asarray(index_array, dtype=np.int64))]


def to_host(tensor):
    return tensor.cpu().numpy()


def ptr(tensor):
    return tensor.data_ptr() if tensor is not None else None


def reduction_buffers():
    """(out[4], scratch[TPA_RED_SCRATCH]) doubles for the deterministic reductions: per device AND per host thread.  (Round 4: one
    pair per device was shared by all threads -- with the reference's ``DMRGThreadPlusHC`` pattern, two threads contracting at the same
    time, a second ``inner`` / ``norm`` overwrote the result of the first before it was read; found by tests/test_threads.py.)"""
    import threading
    t = torch()
    key = (t.cuda.current_device(), threading.get_ident())
    ent = _scratch.get(key)
    if ent is None:
        ent = _scratch[key] = (t.zeros(4, dtype=t.float64, device='cuda'), t.zeros(4096, dtype=t.float64, device='cuda'))
    return ent


def read_scalar(out, cplx):
    """Blocking read of a reduction result written to ``out[0:2]``."""
    v = out[:2].cpu()
    return complex(float(v[0]), float(v[1])) if cplx else float(v[0])


class ScalarPipe:
    """A few device-resident doubles per step of an iteration plus their delayed read-back: the device buffer the kernels
    write into (``dev[step]``, 2 doubles) and a pinned host mirror filled by asynchronous copies, each followed by an event
    on the launch stream.  ``get(step)`` waits for THAT copy only, so the host can be one step behind the device."""

    def __init__(self, n_steps):
        t = torch()
        self.dev = t.zeros((n_steps, 2), dtype=t.float64, device='cuda')
        self.host = t.zeros((n_steps, 2), dtype=t.float64).pin_memory()
        self.events = [None] * n_steps

    def ptr(self, step, which=0):
        return self.dev.data_ptr() + 8 * (2 * step + which)

    def post(self, step):
        t = torch()
        self.host[step].copy_(self.dev[step], non_blocking=True)
        ev = t.cuda.Event()
        ev.record()
        self.events[step] = ev

    def get(self, step):
        self.events[step].synchronize()
        return float(self.host[step, 0]), float(self.host[step, 1])


_budget_cache = {}


def memory_budget(env_name, fraction, fallback_gb):
    """Byte cap for a cache / batch work area: ``env_name`` (GB) if set, else ``fraction`` of the TOTAL memory of the current
    device but at most 0.9 of what is FREE at the first call (``torch.cuda.mem_get_info``), else ``fallback_gb`` when no device can be asked (the emulated device of the CPU
    tests).  The defaults of rounds 3-4 (48 GB of warm-start bases, 96 GB of batched TEBD work areas) were 1/6 and 1/3 of an
    MI355X's 288 GB, hard-coded; a smaller GPU ran out of memory."""
    got = _budget_cache.get(env_name)
    if got is not None:
        return got
    val = os.environ.get(env_name)
    if val is not None:
        nbytes = int(float(val) * (1 << 30))
    else:
        try:
            t = torch()
            if t.cuda.is_available():
                free, total = t.cuda.mem_get_info()
                # (several processes on one GPU -- xdist workers of the reference suite, ranks sharing a device -- must not
                #  each claim a fraction of the WHOLE device: never more than what is free at first use)
                nbytes = int(min(fraction * total, 0.9 * free))
            else:
                nbytes = int(fallback_gb * (1 << 30))
        except Exception:
            nbytes = int(fallback_gb * (1 << 30))
    _budget_cache[env_name] = nbytes
    return nbytes


def check(rc, what=""):
    _lib.check(rc, what)


c_int = ctypes.c_int
byref = ctypes.byref
E_NOCONV, E_NAN = _lib.E_NOCONV, _lib.E_NAN
