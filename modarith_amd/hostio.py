"""Host-resident batches: chunked, double-buffered upload -> kernel -> download through the C ABI alone.

The field path takes device pointers (include/modarith_amd.h); a caller whose elements live in host memory (the
situation of the reference's CPU callers) pays the PCIe link, so the job is to keep the link busy in both
directions while the kernels run: three streams (upload, compute, download), two device slots, chunk i+1 is
uploaded while chunk i is computed and chunk i-1 is downloaded.  Everything here goes through the exported
`modarith_amd_*` utilities and the `<fn>_<PRIME>_batch` entry points with ctypes -- no torch -- so it is also the
reference for a plain-C caller.  Host arrays must be page-locked (`PinnedArray`) for
the copies to be asynchronous.

Measured (round 2, 2^24 elements of 2^255-19, Gen5 x16): 33.5 ms with 2^19-element chunks against 35.5 ms
for upload-all / compute / download-all -- on this platform uploads and downloads do not run concurrently (the two
directions together move ~60 GB/s, one direction's rate), so the gain of pipelining is the bounded device footprint
(two slots of three chunk buffers instead of three full arrays), not time."""
from __future__ import annotations

import ctypes
from ctypes import c_void_p
from typing import Optional

import numpy as np

from . import _lib

BINARY = ("modmul", "modadd", "modsub", "modadd_lazy", "modsub_lazy")
UNARY = ("modsqr", "modneg", "modneg_lazy", "nres", "redc", "modcpy")


class PinnedArray:
    """page-locked host array [N, n] of uint64 (limb-interleaved SoA), owned by the library's allocator"""

    def __init__(self, nlimbs: int, n: int):
        self.lib = _lib.load()
        self.ptr = c_void_p()
        self.nbytes = nlimbs * n * 8
        _lib.check(self.lib.modarith_amd_host_alloc(ctypes.byref(self.ptr), max(self.nbytes, 8)), "host_alloc")
        buf = (ctypes.c_uint64 * (nlimbs * n)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=np.uint64).reshape(nlimbs, n)

    def close(self):
        if self.ptr:
            self.array = None
            _lib.check(self.lib.modarith_amd_host_free(self.ptr), "host_free")
            self.ptr = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def map_host(prime: str, fn: str, a: PinnedArray, b: Optional[PinnedArray], out: PinnedArray, chunk: int = 1 << 19) -> None:
    """out[:, j] = fn(a[:, j], b[:, j]) for host-resident SoA batches, pipelined over the link in chunks of `chunk`
    elements.  fn is one of BINARY (needs b) or UNARY."""
    lib = _lib.load()
    if fn in BINARY:
        if b is None:
            raise ValueError("%s needs two operands" % fn)
    elif fn in UNARY:
        b = None
    else:
        raise ValueError("unsupported function %r" % fn)
    N, n = a.array.shape
    for other in (b, out):
        if other is not None and other.array.shape != (N, n):
            raise ValueError("operands must share the shape [Nlimbs, n]")
    f = getattr(lib, "%s_%s_batch" % (fn, prime))
    if n == 0:
        return
    chunk = max(2, min(chunk, n) & ~1)
    up, run, down = c_void_p(), c_void_p(), c_void_p()
    for s in (up, run, down):
        _lib.check(lib.modarith_amd_stream_create(ctypes.byref(s)), "stream_create")
    nslots = 2
    nbuf = 3 if b is not None else 2
    slots = []
    try:
        for _ in range(nslots):
            bufs = []
            for _ in range(nbuf):
                d = c_void_p()
                _lib.check(lib.modarith_amd_malloc(ctypes.byref(d), N * chunk * 8), "malloc")
                bufs.append(d)
            slots.append(bufs)
        row = n * 8                                    # bytes between limb rows on the host

        def download(off, cnt, dc):
            _lib.check(lib.modarith_amd_stream_wait(down, run), "stream_wait")
            for limb in range(N):
                _lib.check(lib.modarith_amd_memcpy_d2h(out.ptr.value + limb * row + off * 8, dc.value + limb * chunk * 8, cnt * 8, down), "d2h")

        pending = None                                 # (off, cnt, dc) of the chunk whose kernel is enqueued but not its download
        for i, off in enumerate(range(0, n, chunk)):
            cnt = min(chunk, n - off)
            bufs = slots[i % nslots]
            da, dc = bufs[0], bufs[-1]
            db = bufs[1] if b is not None else None
            # slot i%2 was last used by chunk i-2; its download is the newest one on `down` right now (chunk i-1's is
            # enqueued only below), so waiting for `down` here frees the slot without serialising against chunk i-1
            _lib.check(lib.modarith_amd_stream_wait(up, down), "stream_wait")
            for limb in range(N):                      # one copy per limb row: the device chunk has ld = chunk
                _lib.check(lib.modarith_amd_memcpy_h2d(da.value + limb * chunk * 8, a.ptr.value + limb * row + off * 8, cnt * 8, up), "h2d")
                if db is not None:
                    _lib.check(lib.modarith_amd_memcpy_h2d(db.value + limb * chunk * 8, b.ptr.value + limb * row + off * 8, cnt * 8, up), "h2d")
            if pending is not None:
                download(*pending)
            _lib.check(lib.modarith_amd_stream_wait(run, up), "stream_wait")
            if db is not None:
                _lib.check(f(da, db, dc, cnt, chunk, run), fn)
            else:
                _lib.check(f(da, dc, cnt, chunk, run), fn)
            pending = (off, cnt, dc)
        download(*pending)
        _lib.check(lib.modarith_amd_sync(down), "sync")
    finally:
        for s in (up, run, down):
            lib.modarith_amd_sync(s)
        for bufs in slots:
            for d in bufs:
                lib.modarith_amd_free(d)
        for s in (up, run, down):
            lib.modarith_amd_stream_destroy(s)


class PinnedBytes:
    """page-locked host array [n, nbytes] of uint8 (contiguous byte records, the layout of rfc7748's bk / bu / bv:
    simd/rfc7748_simt.cu:165-168), owned by the library's allocator"""

    def __init__(self, n: int, nbytes: int):
        self.lib = _lib.load()
        self.ptr = c_void_p()
        self.nbytes = n * nbytes
        _lib.check(self.lib.modarith_amd_host_alloc(ctypes.byref(self.ptr), max(self.nbytes, 8)), "host_alloc")
        buf = (ctypes.c_uint8 * (n * nbytes)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=np.uint8).reshape(n, nbytes)

    def close(self):
        if self.ptr:
            self.array = None
            _lib.check(self.lib.modarith_amd_host_free(self.ptr), "host_free")
            self.ptr = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ladder_host(curve: str, bk: PinnedBytes, bu: PinnedBytes, bv: PinnedBytes, chunk: int = 1 << 20) -> None:
    """bv[j] = rfc7748(bk[j], bu[j]) for HOST-resident records -- the shape of the reference's own GPU harness
    (simd/rfc7748_simt.cu:257-312: cudaMemcpy in, launch, cudaMemcpy out), pipelined: three streams, two device slots, the
    upload of chunk i+1 and the download of chunk i-1 run under the ladders of chunk i.  The ladder is VALU-bound (2^20
    X25519 records take ~9 ms, their 64 MiB of input ~1.2 ms of link time), so the link disappears behind the kernels.
    C ABI only (rfc7748_<C>_batch_ws with a caller-owned workspace per slot)."""
    lib = _lib.load()
    if curve not in _lib.LADDERS:
        raise ValueError("curve must be one of %s" % (_lib.LADDERS,))
    n, nb = bk.array.shape
    if bu.array.shape != (n, nb) or bv.array.shape != (n, nb):
        raise ValueError("bk, bu, bv must share the shape [n, Nbytes]")
    if n == 0:
        return
    f = getattr(lib, "rfc7748_%s_batch_ws" % curve)
    chunk = max(1, min(chunk, n))
    wsb = int(getattr(lib, "rfc7748_%s_batch_workspace_bytes" % curve)(chunk))
    up, run, down = c_void_p(), c_void_p(), c_void_p()
    for s in (up, run, down):
        _lib.check(lib.modarith_amd_stream_create(ctypes.byref(s)), "stream_create")
    slots = []
    try:
        for _ in range(2):
            bufs = []
            for size in (chunk * nb, chunk * nb, chunk * nb, max(wsb, 8)):
                d = c_void_p()
                _lib.check(lib.modarith_amd_malloc(ctypes.byref(d), size), "malloc")
                bufs.append(d)
            slots.append(bufs)

        def download(off, cnt, dv):
            _lib.check(lib.modarith_amd_stream_wait(down, run), "stream_wait")
            _lib.check(lib.modarith_amd_memcpy_d2h(bv.ptr.value + off * nb, dv, cnt * nb, down), "d2h")

        pending = None
        for i, off in enumerate(range(0, n, chunk)):
            cnt = min(chunk, n - off)
            dk, du, dv, ws = slots[i % 2]
            _lib.check(lib.modarith_amd_stream_wait(up, down), "stream_wait")      # slot i%2: chunk i-2's download is the newest on `down`
            _lib.check(lib.modarith_amd_memcpy_h2d(dk, bk.ptr.value + off * nb, cnt * nb, up), "h2d")
            _lib.check(lib.modarith_amd_memcpy_h2d(du, bu.ptr.value + off * nb, cnt * nb, up), "h2d")
            if pending is not None:
                download(*pending)
            _lib.check(lib.modarith_amd_stream_wait(run, up), "stream_wait")
            _lib.check(f(dk, du, dv, cnt, ws, max(wsb, 8), run), "rfc7748_%s_batch_ws" % curve)
            pending = (off, cnt, dv)
        download(*pending)
        _lib.check(lib.modarith_amd_sync(down), "sync")
    finally:
        for s in (up, run, down):
            lib.modarith_amd_sync(s)
        for bufs in slots:
            for d in bufs:
                lib.modarith_amd_free(d)
        for s in (up, run, down):
            lib.modarith_amd_stream_destroy(s)
