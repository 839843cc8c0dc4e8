"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_int, c_long, c_size_t, c_uint, c_uint64, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
U64P = POINTER(c_uint64)
INTP = POINTER(c_int)

PRIMES = {"X25519": (5, 51, 255, 32), "NIST256": (5, 52, 256, 32), "X448": (8, 56, 448, 56)}
# primes whose field.c function set is the generic oracle bound to the constants captured from the reference
# (oracle/field_<P>.c with oracle_bind_<P>): they carry a curve-layer oracle
BOUND_PRIMES = {"NIST384": (7, 56, 384, 48), "NIST521": (9, 58, 521, 66), "SECP256K1": (5, 52, 256, 32), "NUMS256W": (5, 52, 256, 32),
                "ED248": (5, 51, 251, 32), "ED376": (7, 55, 383, 48), "ED500": (9, 57, 505, 64)}
ORACLE_CURVES = (("ed25519", "X25519"), ("ed448", "X448"), ("nist256", "NIST256"), ("nist384", "NIST384"), ("nist521", "NIST521"),
                 ("secp256k1", "SECP256K1"), ("nums256w", "NUMS256W"), ("nums256e", "NUMS256W"),
                 ("ed248", "ED248"), ("ed376", "ED376"), ("ed500", "ED500"))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ODIR, "all"])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        self._bound = {}
        for P in BOUND_PRIMES:
            from tests.generic_oracle import params_from_golden
            self._bound[P] = params_from_golden(P)             # kept alive; the C side copies it
            bind = getattr(lib, "oracle_bind_" + P)
            bind.argtypes = [c_void_p]; bind.restype = c_int
            assert bind(ctypes.byref(self._bound[P])) == 0, "parameter block does not match oracle/field_%s.c" % P
        allp = dict(PRIMES); allp.update(BOUND_PRIMES)
        self.primes = allp
        for P in allp:
            g = lambda f: getattr(lib, "%s_%s" % (f, P))
            for f in ("modadd", "modsub", "modmul", "modadd_lazy", "modsub_lazy"):
                g(f).argtypes = [U64P, U64P, U64P]; g(f).restype = None
            for f in ("modneg", "modneg_lazy", "modsqr", "modcpy", "nres", "redc", "modpro"):
                g(f).argtypes = [U64P, U64P]; g(f).restype = None
            g("modinv").argtypes = [U64P, U64P, U64P]; g("modinv").restype = None
            g("modsqrt").argtypes = [U64P, U64P, U64P]; g("modsqrt").restype = None
            g("modqr").argtypes = [U64P, U64P]; g("modqr").restype = c_int
            g("batch_modsqrt").argtypes = [c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modsqrt").restype = None
            g("batch_modqr").argtypes = [c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modqr").restype = None
            g("modmli").argtypes = [U64P, c_int, U64P]; g("modmli").restype = None
            g("modnsqr").argtypes = [U64P, c_int]; g("modnsqr").restype = None
            for f in ("flatten", "modfsb"):
                g(f).argtypes = [U64P]; g(f).restype = c_uint64
            for f in ("modis1", "modis0", "modsign"):
                g(f).argtypes = [U64P]; g(f).restype = c_int
            g("modcmp").argtypes = [U64P, U64P]; g("modcmp").restype = c_int
            for f in ("modzer", "modone", "modhaf"):
                g(f).argtypes = [U64P]; g(f).restype = None
            g("modint").argtypes = [c_int, U64P]; g("modint").restype = None
            g("mod2r").argtypes = [c_uint, U64P]; g("mod2r").restype = None
            g("modcmv").argtypes = [c_int, U64P, U64P]; g("modcmv").restype = None
            g("modcsw").argtypes = [c_int, U64P, U64P]; g("modcsw").restype = None
            g("modshl").argtypes = [c_uint, U64P]; g("modshl").restype = None
            g("modshr").argtypes = [c_uint, U64P]; g("modshr").restype = c_int
            g("modexp").argtypes = [U64P, ctypes.c_char_p]; g("modexp").restype = None
            g("modimp").argtypes = [ctypes.c_char_p, U64P]; g("modimp").restype = c_int
            g("time_modmul").argtypes = [U64P, U64P, c_long]; g("time_modmul").restype = c_uint
            g("time_modsqr").argtypes = [U64P, c_long]; g("time_modsqr").restype = c_uint
            g("time_modinv").argtypes = [U64P, c_long]; g("time_modinv").restype = c_uint
            for f in ("batch_modmul", "batch_modadd", "batch_modsub", "batch_modadd_lazy", "batch_modsub_lazy", "batch_modmul_shared"):
                g(f).argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]; g(f).restype = None
            for f in ("batch_modsqr", "batch_modneg", "batch_modneg_lazy", "batch_nres", "batch_redc", "batch_modcpy", "batch_modinv"):
                g(f).argtypes = [c_void_p, c_void_p, c_size_t, c_size_t]; g(f).restype = None
            g("batch_modmli").argtypes = [c_void_p, c_int, c_void_p, c_size_t, c_size_t]; g("batch_modmli").restype = None
            g("batch_modcsw").argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modcsw").restype = None
            g("batch_modcmv").argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modcmv").restype = None
            g("batch_modfsb").argtypes = [c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modfsb").restype = None
            g("batch_modimp").argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modimp").restype = None
            g("batch_modexp").argtypes = [c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_modexp").restype = None
        for C in ("X25519", "X448"):
            f = getattr(lib, "rfc7748_" + C)
            f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]; f.restype = None
            f = getattr(lib, "batch_rfc7748_" + C)
            f.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]; f.restype = None
        # Edwards layer (oracle/edwards_oracle.c): point = struct {x[NL], y[NL], z[NL]}
        self.ed = {}
        for C, P in ORACLE_CURVES:
            nl = allp[P][0]

            class Pt(ctypes.Structure):
                _fields_ = [("x", c_uint64 * nl), ("y", c_uint64 * nl), ("z", c_uint64 * nl)]
            PP = POINTER(Pt)
            g = lambda f: getattr(lib, "ecn_%s_%s" % (C, f))
            sig = {"cpy": [PP, PP], "neg": [PP], "add": [PP, PP], "sub": [PP, PP], "dbl": [PP], "inf": [PP], "affine": [PP],
                   "cof": [PP], "gen": [PP], "ran": [c_int, PP], "mul": [ctypes.c_char_p, PP],
                   "mul2": [ctypes.c_char_p, PP, ctypes.c_char_p, PP, PP], "set": [c_int, ctypes.c_char_p, ctypes.c_char_p, PP]}
            for f, a in sig.items():
                g(f).argtypes = a; g(f).restype = None
            g("isinf").argtypes = [PP]; g("isinf").restype = c_int
            g("cmp").argtypes = [PP, PP]; g("cmp").restype = c_int
            g("get").argtypes = [PP, ctypes.c_char_p, ctypes.c_char_p]; g("get").restype = c_int
            g("batch_mul").argtypes = [c_void_p, c_void_p, c_size_t, c_size_t]; g("batch_mul").restype = None
            self.ed[C] = (Pt, allp[P][3])
        lib.oracle_parallel.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int]
        lib.oracle_parallel.restype = c_int

    def fn(self, name, prime):
        return getattr(self.lib, "%s_%s" % (name, prime))

    @staticmethod
    def arr(prime, vals=None):
        n = (PRIMES.get(prime) or BOUND_PRIMES[prime])[0]
        return (c_uint64 * n)(*(vals if vals is not None else [0] * n))

    # element-level helpers returning python lists
    def un(self, f, P, a):
        z = self.arr(P)
        self.fn(f, P)(self.arr(P, a), z)
        return list(z)

    def bi(self, f, P, a, b):
        z = self.arr(P)
        self.fn(f, P)(self.arr(P, a), self.arr(P, b), z)
        return list(z)

    # ---- Edwards helpers (C = "ed25519" | "ed448")
    def ecn(self, C, f):
        return getattr(self.lib, "ecn_%s_%s" % (C, f))

    def ed_point(self, C, x_hex=None, y_hex=None):
        Pt, nb = self.ed[C]
        p = Pt()
        if x_hex is None:
            self.ecn(C, "inf")(ctypes.byref(p))
        else:
            self.ecn(C, "set")(0, bytes.fromhex(x_hex), bytes.fromhex(y_hex), ctypes.byref(p))
        return p

    def ed_xy(self, C, p):
        """affine (x, y) hex of a copy of p"""
        Pt, nb = self.ed[C]
        q = Pt()
        self.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(q))
        x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        self.ecn(C, "get")(ctypes.byref(q), x, y)
        return [x.raw.hex(), y.raw.hex()]

    def ladder(self, curve, k, u):
        nb = PRIMES[curve][3]
        out = ctypes.create_string_buffer(nb)
        getattr(self.lib, "rfc7748_" + curve)(bytes(k), bytes(u), out)
        return out.raw


def load_oracle(build=True):
    so = os.path.join(ODIR, "liboracle.so")
    if build or not os.path.exists(so):
        build_oracle()
    return Oracle(ctypes.CDLL(so))
