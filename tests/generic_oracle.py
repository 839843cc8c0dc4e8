"""ctypes view of oracle/field_generic.c: a run-time interpretation of the reference generators' algorithms,
parameterised from the constants captured from the reference (tests/golden/field_<P>.json "params").
TEST INFRASTRUCTURE ONLY."""
import ctypes
from ctypes import c_int, c_longlong, c_size_t, c_uint64, c_void_p, POINTER

from tests.conftest import load_golden

GMAXN = 16
U64P = POINTER(c_uint64)


class GParams(ctypes.Structure):
    _fields_ = [("family", c_int), ("n", c_int), ("radix", c_int), ("nbits", c_int), ("nbytes", c_int), ("xcess", c_int),
                ("pm1d2", c_int), ("m", c_uint64), ("mm", c_uint64), ("epm", c_int), ("fred", c_int), ("carry_on", c_int),
                ("ppw", c_longlong * (GMAXN + 1)), ("E", c_int), ("trin", c_int), ("neg_limb", c_int),
                ("ndash", c_uint64), ("barrett_r", c_uint64), ("r2", c_uint64 * GMAXN),
                ("pp_cnt", c_int), ("pp_idx", c_int * GMAXN), ("pp_sgn", c_int * GMAXN), ("pp_val", c_uint64 * GMAXN),
                ("pe_words", c_int), ("pe", c_uint64 * GMAXN), ("roi", c_uint64 * GMAXN), ("overflow", c_int), ("bad_overflow", c_int)]


def _i(v):
    return int(v, 16) if isinstance(v, str) else int(v)


def params_from_golden(prime):
    """fill the parameter block from what the reference generator printed / derived for this prime"""
    g = load_golden("field_%s.json" % prime)["params"]
    P = GParams()
    N, radix, p = g["N"], g["base"], _i(g["p"])
    P.n, P.radix, P.nbits, P.nbytes, P.xcess, P.pm1d2 = N, radix, g["n"], g["Nbytes"], g["xcess"], g["PM1D2"]
    pe = _i(g["PE"])
    words = [(pe >> (64 * i)) & (2**64 - 1) for i in range((pe.bit_length() + 63) // 64)]
    P.pe_words = len(words)
    for i, w in enumerate(words):
        P.pe[i] = w
    for i, v in enumerate(g["ROI"]):
        P.roi[i] = _i(v)
    if "ppw" in g:                                           # monty.py
        P.family = 1
        ppw = [(-_i(v[1:]) if v.startswith("-") else _i(v)) for v in g["ppw"]]
        for i, v in enumerate(ppw):
            P.ppw[i] = v
        P.E, P.trin, P.ndash = int(g["E"]), g["trin"], _i(g["ndash"])
        neg = [i for i, v in enumerate(ppw) if i > 0 and v == -1]
        P.neg_limb = neg[0] if neg else 0
        br = (1 << (g["n"] + radix)) // p
        P.barrett_r = br if br < 2**64 else 0
        for i, v in enumerate(g["cw"]):
            P.r2[i] = _i(v)
        pp = [(i, -1 if v < 0 else 1, abs(v)) for i, v in enumerate(ppw[:N]) if v]
        if P.E:
            pp = [t for t in pp if t[0] != N - 1] + [(N - 1, 1, ppw[N - 1] + (1 << radix))]
    else:                                                    # pseudo.py
        P.family = 0
        P.m, P.mm = _i(g["m"]), _i(g["mm"])
        P.epm, P.fred, P.carry_on = int(g["EPM"]), int(g["fred"]), int(g["carry_on"])
        P.overflow = int(g["overflow"])
        assert g["bad_overflow_mul"] == g["bad_overflow_sqr"], "the generators set both flags together at 64 bits (no Karatsuba)"
        P.bad_overflow = int(g["bad_overflow_mul"])
        
        pp = [(0, -1, _i(g["m"])), (N - 1, 1, _i(g["TW"]))]
    P.pp_cnt = len(pp)
    for k, (i, s, v) in enumerate(pp):
        P.pp_idx[k], P.pp_sgn[k], P.pp_val[k] = i, s, v
    return P


class Generic:
    def __init__(self, lib, prime):
        self.lib, self.P, self.N = lib, params_from_golden(prime), load_golden("field_%s.json" % prime)["params"]["N"]
        assert lib.gen_params_size() == ctypes.sizeof(GParams)
        R = ctypes.byref(self.P)
        self.R = R
        L = lib
        for f in ("gen_modadd", "gen_modsub", "gen_modmul"):
            getattr(L, f).argtypes = [c_void_p, U64P, U64P, U64P]; getattr(L, f).restype = None
        for f in ("gen_modneg", "gen_modsqr", "gen_nres", "gen_redc", "gen_modpro"):
            getattr(L, f).argtypes = [c_void_p, U64P, U64P]; getattr(L, f).restype = None
        L.gen_modmli.argtypes = [c_void_p, U64P, c_int, U64P]; L.gen_modmli.restype = None
        L.gen_modinv.argtypes = [c_void_p, U64P, U64P, U64P]; L.gen_modinv.restype = None
        L.gen_modsqrt.argtypes = [c_void_p, U64P, U64P, U64P]; L.gen_modsqrt.restype = None
        L.gen_modqr.argtypes = [c_void_p, U64P, U64P]; L.gen_modqr.restype = c_int
        for f in ("gen_modfsb", "gen_flatten"):
            getattr(L, f).argtypes = [c_void_p, U64P]; getattr(L, f).restype = c_uint64
        L.gen_batch.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]; L.gen_batch.restype = None
        L.gen_batch_mli.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_size_t]; L.gen_batch_mli.restype = None

    def arr(self, vals=None):
        return (c_uint64 * self.N)(*(vals if vals is not None else [0] * self.N))

    def bi(self, f, a, b):
        z = self.arr()
        getattr(self.lib, "gen_" + f)(self.R, self.arr(a), self.arr(b), z)
        return list(z)

    def un(self, f, a):
        z = self.arr()
        getattr(self.lib, "gen_" + f)(self.R, self.arr(a), z)
        return list(z)
