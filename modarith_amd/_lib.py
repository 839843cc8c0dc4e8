"""ctypes loader for libmodarith_amd.so (the C-ABI of include/modarith_amd.h).

The reference drives its generated code the same way: compile to a shared object, load it with
ctypes.CDLL and declare argtypes (pseudo.py:1702-1750).  There is NO fallback: if the HIP library
has not been built, loading raises; nothing in the product path computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_size_t, c_uint, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MA_LIB", os.path.join(HERE, "libmodarith_amd.so"))   # MA_LIB: a variant build under test
PRIMES = ("X25519", "NIST256", "X448",
          "NIST521", "PM266", "PM383", "NUMS256W", "NIST384", "NIST224", "SECP256K1M", "NIST256Q", "ED25519Q", "ED448Q",
          "C2065", "PM336", "PM512", "GM270", "GM240", "GM360", "GM480", "GM384", "GM512", "TWEEDLE", "SIDH434", "SIDH503",
          "SECP256K1", "C41417", "ED248", "ED376", "ED500",
          "SIDH610", "SIDH751", "MFP4", "MFP7", "MFP1973", "CSIDH512", "GM378",
          "PM383M", "PM266M", "PM336M", "C41417M", "PM512M", "M607")
LADDERS = ("X25519", "X448")
CURVES = {"ed25519": (5, 32), "ed448": (8, 56), "nist256": (5, 32), "nist384": (7, 48), "nist521": (9, 66), "secp256k1": (5, 32), "nums256w": (5, 32), "nums256e": (5, 32), "ed248": (5, 32), "ed376": (7, 48), "ed500": (9, 64)}       # curve -> (Nlimbs, Nbytes)
ED_BATCH_FUNCS = ("mul", "mul2", "mul2_exact", "ran", "add", "sub", "cpy", "dbl", "neg", "inf", "gen", "cof", "affine", "cmp", "isinf", "set", "get")
FUSED_CURVES = ("ed25519", "ed448", "nist256", "secp256k1")       # fused mul + get kernels (csrc/ed26.h, csrc/ed28.h, csrc/wn26.h)
FUSED2_CURVES = ("ed25519", "ed448", "nist256", "secp256k1")        # fused mul2 + get
FUSEDG_CURVES = ("nist256", "secp256k1", "ed25519", "ed448")             # fused gen + mul + get (fixed-base tables; csrc/wn26.h, ed26.h, ed28.h *_mulgen_get_one)
FUSEDG2_CURVES = ("nist256", "secp256k1", "ed25519", "ed448")    # fused gen + mul2 + get (e*G + f*Q, verification)
FUSED_FUNCS = (("rfc7748_X25519_base_batch", "rfc7748_X448_base_batch") + tuple("ecn_%s_mulgen_get_batch" % c for c in FUSEDG_CURVES)
               + tuple("ecn_%s_mulgen2_get_%s" % (c, f) for c in FUSEDG2_CURVES for f in ("batch", "workspace_bytes")) + tuple("ecn_%s_mul_get_%s" % (c, f) for c in FUSED_CURVES for f in ("batch", "workspace_bytes"))
               + tuple("ecn_%s_mul2_get_%s" % (c, f) for c in FUSED2_CURVES for f in ("batch", "workspace_bytes")))
ED_SCALAR_FUNCS = ("mul2", "ran", "get", "set", "inf", "isinf", "neg", "add", "sub", "dbl", "gen", "mul", "cmp", "affine", "cpy", "cof",
                   "mul_workspace_bytes")

_lib = None

_P = c_void_p
_SIG = {
    # name: argtypes of the _batch form (device pointers as void*)
    "modadd": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modsub": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modadd_lazy": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modsub_lazy": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modmul": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modmuls": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modneg": [_P, _P, c_size_t, c_size_t, _P],
    "modneg_lazy": [_P, _P, c_size_t, c_size_t, _P],
    "modsqr": [_P, _P, c_size_t, c_size_t, _P],
    "modcpy": [_P, _P, c_size_t, c_size_t, _P],
    "modpro": [_P, _P, c_size_t, c_size_t, _P],
    "nres": [_P, _P, c_size_t, c_size_t, _P],
    "redc": [_P, _P, c_size_t, c_size_t, _P],
    "modinv": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modsqrt": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modqr": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modmli": [_P, c_int, _P, c_size_t, c_size_t, _P],
    "modnsqr": [_P, c_int, c_size_t, c_size_t, _P],
    "modfsb": [_P, _P, c_size_t, c_size_t, _P],
    "flatten": [_P, _P, c_size_t, c_size_t, _P],
    "prop": [_P, _P, c_size_t, c_size_t, _P],
    "modhaf": [_P, c_size_t, c_size_t, _P],
    "modshl": [c_uint, _P, c_size_t, c_size_t, _P],
    "modshr": [c_uint, _P, _P, c_size_t, c_size_t, _P],
    "modis1": [_P, _P, c_size_t, c_size_t, _P],
    "modis0": [_P, _P, c_size_t, c_size_t, _P],
    "modsign": [_P, _P, c_size_t, c_size_t, _P],
    "modlimbs": [_P, _P, c_size_t, c_size_t, _P],
    "modcmp": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modzer": [_P, c_size_t, c_size_t, _P],
    "modone": [_P, c_size_t, c_size_t, _P],
    "modint": [c_int, _P, c_size_t, c_size_t, _P],
    "mod2r": [c_uint, _P, c_size_t, c_size_t, _P],
    "modcmv": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modcsw": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modimp": [_P, _P, _P, c_size_t, c_size_t, _P],
    "modexp": [_P, _P, c_size_t, c_size_t, _P],
    "time_protocol": [c_int, _P, _P, _P, ctypes.c_long, c_size_t, c_size_t, _P],
    "moduniform": [ctypes.c_ulonglong, ctypes.c_ulonglong, c_size_t, c_int, _P, c_size_t, c_size_t, _P],
}
BATCH_FUNCS = tuple(_SIG)
# scalar (_ct) names declared by the header, for the symbol-export test
SCALAR_FUNCS = ("prop", "flatten", "modfsb", "modadd", "modsub", "modneg", "modmli", "modmul", "modsqr", "modcpy", "modnsqr",
                "modpro", "modinv", "modqr", "modsqrt", "nres", "redc", "modis1", "modis0", "modzer", "modone", "modint", "modcmv",
                "modcsw", "modshl", "modshr", "modhaf", "mod2r", "modexp", "modimp", "modsign", "modcmp")
UTIL_FUNCS = ("modarith_amd_abi_version", "modarith_amd_last_error", "modarith_amd_device_count",
              "modarith_amd_set_device", "modarith_amd_malloc", "modarith_amd_free", "modarith_amd_memcpy_h2d",
              "modarith_amd_memcpy_d2h", "modarith_amd_sync", "modarith_amd_aos_to_soa", "modarith_amd_soa_to_aos",
              "modarith_amd_stream_create", "modarith_amd_stream_destroy", "modarith_amd_stream_wait", "modarith_amd_host_alloc", "modarith_amd_host_free",
              "modarith_amd_field_info", "modarith_amd_recommended_ld", "modarith_amd_recommended_ld_for", "modarith_amd_batch_words", "modarith_amd_scratch_trim",
              "modarith_amd_last_launch", "modarith_amd_status", "modarith_amd_clear_status", "modarith_amd_thread_status", "modarith_amd_clear_thread_status", "modarith_amd_sclk_probe", "modarith_amd_wall_clock_khz")


def _declare_curve(lib, C: str) -> None:
    """argtypes / restypes of the batched curve entry points ecn_<C>_*_batch of `lib` (the main library or a curve plug-in)"""
    g = lambda f: getattr(lib, "ecn_%s_%s" % (C, f))
    g("mul_workspace_bytes").argtypes = [c_size_t]
    g("mul_workspace_bytes").restype = c_size_t
    g("mul_batch").argtypes = [_P, _P, c_size_t, c_size_t, _P, c_size_t, _P]
    g("mul2_batch").argtypes = [_P, _P, _P, _P, _P, c_size_t, c_size_t, _P, c_size_t, _P]
    g("mul2_exact_batch").argtypes = [_P, _P, _P, _P, _P, c_size_t, c_size_t, _P, c_size_t, _P]
    g("ran_batch").argtypes = [c_int, _P, c_size_t, c_size_t, _P]
    for f in ("add", "sub", "cpy"):
        g(f + "_batch").argtypes = [_P, _P, c_size_t, c_size_t, _P]
    for f in ("dbl", "neg", "inf", "gen", "cof", "affine"):
        g(f + "_batch").argtypes = [_P, c_size_t, c_size_t, _P]
    g("cmp_batch").argtypes = [_P, _P, _P, c_size_t, c_size_t, _P]
    g("isinf_batch").argtypes = [_P, _P, c_size_t, c_size_t, _P]
    g("set_batch").argtypes = [_P, _P, _P, _P, c_size_t, c_size_t, _P]
    g("get_batch").argtypes = [_P, _P, _P, _P, c_size_t, c_size_t, _P]
    for f in ED_BATCH_FUNCS:
        g(f + "_batch").restype = c_int


def load() -> ctypes.CDLL:
    """Load the HIP library or raise: the engine has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "modarith_amd: %s is missing -- build it with `python -m modarith_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        # torch-rocm ships its own HIP runtime: when this process is going to use torch as well, that runtime must be the first one
        # loaded -- with the library's copy first, HIP calls made through the library report "no ROCm-capable device"
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for P in PRIMES:
        for fn, args in _SIG.items():
            f = getattr(lib, "%s_%s_batch" % (fn, P))
            f.argtypes = args
            f.restype = c_int
    for C in LADDERS:
        f = getattr(lib, "rfc7748_%s_batch" % C)
        f.argtypes = [_P, _P, _P, c_size_t, _P]
        f.restype = c_int
        g = getattr(lib, "rfc7748_%s" % C)
        g.argtypes = [c_char_p, c_char_p, c_char_p]
        g.restype = None
        w = getattr(lib, "rfc7748_%s_batch_workspace_bytes" % C)
        w.argtypes = [c_size_t]
        w.restype = c_size_t
        h = getattr(lib, "rfc7748_%s_batch_ws" % C)
        h.argtypes = [_P, _P, _P, c_size_t, _P, c_size_t, _P]
        h.restype = c_int
    for C in CURVES:
        _declare_curve(lib, C)
    for c in FUSED_CURVES:
        f = getattr(lib, "ecn_%s_mul_get_batch" % c)
        f.argtypes = [_P, _P, _P, _P, _P, c_size_t, c_size_t, _P, c_size_t, _P]
        f.restype = c_int
        f = getattr(lib, "ecn_%s_mul_get_workspace_bytes" % c)
        f.argtypes = [c_size_t]
        f.restype = c_size_t
    for c in FUSEDG_CURVES:
        f = getattr(lib, "ecn_%s_mulgen_get_batch" % c)
        f.argtypes = [_P, _P, _P, _P, c_size_t, _P]
        f.restype = c_int
    for C in LADDERS:
        f = getattr(lib, "rfc7748_%s_base_batch" % C)
        f.argtypes = [_P, _P, c_size_t, _P]
        f.restype = c_int
    for c in FUSEDG2_CURVES:
        f = getattr(lib, "ecn_%s_mulgen2_get_batch" % c)
        f.argtypes = [_P, _P, _P, _P, _P, _P, c_size_t, c_size_t, _P, c_size_t, _P]
        f.restype = c_int
        f = getattr(lib, "ecn_%s_mulgen2_get_workspace_bytes" % c)
        f.argtypes = [c_size_t]
        f.restype = c_size_t
    for c in FUSED2_CURVES:
        f = getattr(lib, "ecn_%s_mul2_get_batch" % c)
        f.argtypes = [_P, _P, _P, _P, _P, _P, _P, c_size_t, c_size_t, _P, c_size_t, _P]
        f.restype = c_int
        f = getattr(lib, "ecn_%s_mul2_get_workspace_bytes" % c)
        f.argtypes = [c_size_t]
        f.restype = c_size_t
    lib.modarith_amd_last_error.restype = c_char_p
    lib.modarith_amd_abi_version.restype = c_int
    lib.modarith_amd_device_count.restype = c_int
    lib.modarith_amd_aos_to_soa.argtypes = [_P, _P, c_size_t, c_int, c_size_t, _P]
    lib.modarith_amd_soa_to_aos.argtypes = [_P, _P, c_size_t, c_int, c_size_t, _P]
    lib.modarith_amd_field_info.argtypes = [c_char_p] + [ctypes.POINTER(c_int)] * 5
    lib.modarith_amd_malloc.argtypes = [ctypes.POINTER(c_void_p), c_size_t]
    lib.modarith_amd_free.argtypes = [_P]
    lib.modarith_amd_memcpy_h2d.argtypes = [_P, _P, c_size_t, _P]
    lib.modarith_amd_memcpy_d2h.argtypes = [_P, _P, c_size_t, _P]
    lib.modarith_amd_sync.argtypes = [_P]
    lib.modarith_amd_stream_create.argtypes = [ctypes.POINTER(c_void_p)]
    lib.modarith_amd_stream_destroy.argtypes = [_P]
    lib.modarith_amd_stream_wait.argtypes = [_P, _P]
    lib.modarith_amd_host_alloc.argtypes = [ctypes.POINTER(c_void_p), c_size_t]
    lib.modarith_amd_host_free.argtypes = [_P]
    lib.modarith_amd_recommended_ld.argtypes = [c_size_t]
    lib.modarith_amd_recommended_ld.restype = c_size_t
    lib.modarith_amd_recommended_ld_for.argtypes = [c_size_t, c_int]
    lib.modarith_amd_recommended_ld_for.restype = c_size_t
    lib.modarith_amd_batch_words.argtypes = [c_size_t, c_int, c_size_t]
    lib.modarith_amd_batch_words.restype = c_size_t
    lib.modarith_amd_scratch_trim.argtypes = [c_size_t]
    lib.modarith_amd_scratch_trim.restype = c_int
    lib.modarith_amd_last_launch.restype = c_char_p
    lib.modarith_amd_status.restype = c_int
    lib.modarith_amd_clear_status.restype = None
    lib.modarith_amd_sclk_probe.argtypes = [_P, ctypes.c_uint, ctypes.c_uint, _P]
    lib.modarith_amd_sclk_probe.restype = c_int
    lib.modarith_amd_wall_clock_khz.restype = c_int
    _lib = lib
    return lib


_plugins = {}


def load_plugin(tag: str, path: str = None) -> ctypes.CDLL:
    """Load the plug-in of a generated field (modarith_amd.generate): same entry points as a built-in prime, under
    <fn>_<tag>_batch / <fn>_<tag>_ct.  The main library is loaded first; the plug-in's DT_NEEDED entry resolves to it."""
    if tag in _plugins:
        return _plugins[tag]
    load()
    if path is None:
        from .generate import plugin_path
        path = plugin_path(tag)
    if not os.path.exists(path):
        raise RuntimeError("modarith_amd: no plug-in for %r (%s) -- generate it with `python -m modarith_amd.generate 64 <prime>`. "
                           "There is no CPU fallback." % (tag, path))
    lib = ctypes.CDLL(path)
    for fn, args in _SIG.items():
        f = getattr(lib, "%s_%s_batch" % (fn, tag))
        f.argtypes = args
        f.restype = c_int
    _plugins[tag] = lib
    return lib


_curve_plugins = {}


def load_curve_plugin(name: str, path: str = None):
    """Load the plug-in of a generated curve (modarith_amd.generate.generate_curve); returns (CDLL, Nlimbs, Nbytes)"""
    low = name.lower()
    if low in _curve_plugins:
        return _curve_plugins[low]
    load()
    import json
    from .generate import PLUGIN_DIR, curve_plugin_path
    path = path or curve_plugin_path(low)
    meta = os.path.join(os.path.dirname(path), "curve_%s.json" % name.upper())
    if not (os.path.exists(path) and os.path.exists(meta)):
        raise RuntimeError("modarith_amd: no plug-in for curve %r (%s) -- generate it with modarith_amd.generate.generate_curve(...). "
                           "There is no CPU fallback." % (name, path))
    m = json.load(open(meta))
    field = m["field"]
    if field not in PRIMES:
        load_plugin(field)                 # the curve's field is itself a plug-in
    lib = ctypes.CDLL(path)
    _declare_curve(lib, low)
    _curve_plugins[low] = (lib, m["nlimbs"], m["nbytes"])
    return _curve_plugins[low]


class DeviceError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = load().modarith_amd_last_error().decode(errors="replace")
        raise DeviceError("%s failed (hipError %d): %s" % (what, rc, msg))
