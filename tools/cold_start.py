#!/usr/bin/env python3
"""round 5: what the first call costs.  A fresh process, ctypes and the C ABI only (no torch): dlopen of libmodarith_amd.so (62 MB,
~3 900 kernels in 89 code objects), the first device call, the first launch of a kernel of one prime (the runtime loads that code
object on first use), a second launch, then the first launch of other units.  Run it several times: the first run after a box comes up
also pays the page-in of the file."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "modarith_amd", "libmodarith_amd.so")
t = time.perf_counter()
L = ctypes.CDLL(path)
t_dl = time.perf_counter() - t
L.modarith_amd_last_error.restype = ctypes.c_char_p
vp, sz = ctypes.c_void_p, ctypes.c_size_t
t = time.perf_counter()
ndev = L.modarith_amd_device_count()
t_dev = time.perf_counter() - t
L.modarith_amd_malloc.argtypes = [ctypes.POINTER(vp), sz]
L.modarith_amd_sync.argtypes = [vp]


def buf(nbytes):
    p = vp()
    assert L.modarith_amd_malloc(ctypes.byref(p), nbytes) == 0, L.modarith_amd_last_error()
    return p


def timed(fn):
    t0 = time.perf_counter()
    rc = fn()
    L.modarith_amd_sync(None)
    assert rc == 0, L.modarith_amd_last_error()
    return (time.perf_counter() - t0) * 1e3


n = 4096
t = time.perf_counter()
a, b, c = buf(n * 8 * 8), buf(n * 8 * 8), buf(n * 8 * 8)
t_alloc = (time.perf_counter() - t) * 1e3
out = {"dlopen_ms": t_dl * 1e3, "first_device_call_ms": t_dev * 1e3, "devices": ndev, "first_malloc_ms": t_alloc}
for P in ("X25519", "NIST256", "X448", "SIDH751"):
    f = getattr(L, "modzer_%s_batch" % P)           # fills a batch: safe inputs for the product whatever the buffers held
    f.argtypes = [vp, sz, sz, vp]
    g = getattr(L, "modmul_%s_batch" % P)
    g.argtypes = [vp, vp, vp, sz, sz, vp]
    out["first_call_%s_ms" % P] = timed(lambda: f(a, n, n, None) or f(b, n, n, None) or g(a, b, c, n, n, None))
    out["second_call_%s_ms" % P] = timed(lambda: g(a, b, c, n, n, None))
k = buf(n * 32)
L.rfc7748_X25519_batch.argtypes = [vp, vp, vp, sz, vp]
out["first_rfc7748_X25519_ms"] = timed(lambda: L.rfc7748_X25519_batch(k, k, c, n, None))
out["second_rfc7748_X25519_ms"] = timed(lambda: L.rfc7748_X25519_batch(k, k, c, n, None))
L.ecn_ed25519_gen_batch.argtypes = [vp, sz, sz, vp]
L.ecn_ed25519_mul_get_batch.argtypes = [vp, vp, vp, vp, vp, sz, sz, vp, sz, vp]
pts = buf(n * 15 * 8)
out["first_ecn_ed25519_gen_ms"] = timed(lambda: L.ecn_ed25519_gen_batch(pts, n, n, None))
out["first_ecn_ed25519_mul_get_ms"] = timed(lambda: L.ecn_ed25519_mul_get_batch(k, pts, a, b, None, n, n, None, 0, None))
out["second_ecn_ed25519_mul_get_ms"] = timed(lambda: L.ecn_ed25519_mul_get_batch(k, pts, a, b, None, n, n, None, 0, None))
print(" ".join("%s=%.1f" % (k_, v) if isinstance(v, float) else "%s=%s" % (k_, v) for k_, v in out.items()))
