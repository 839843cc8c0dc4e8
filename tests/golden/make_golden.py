#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

  python tests/golden/make_golden.py            # rewrites tests/golden/*.json

Source of truth (see refgen.py for how the generators are driven without any stand-in tool):
  * field functions: the C text emitted by /root/reference/pseudo.py (X25519) and monty.py
    (NIST256, X448) for a 64-bit word, compiled with gcc and called through ctypes exactly
    as the reference's own self-test does (pseudo.py:1702-1750).  generic=True (default) and,
    for the ladder primes, generic=False (rfc7748.c:20) variants of modadd/modsub/modneg.
  * time.c protocol check words (pseudo.py:1235-1250, 1306-1318): the reference-emitted
    modmul/modsqr run in the reference's loop shape by a small harness appended to the emitted C.
  * modpro/modinv/modsqrt/modqr are NOT emitted (they need the external `addchain` tool);
    their fixtures are the mathematical values after redc (chain-independent), as SURVEY 8(c)
    caveat (1) prescribes.
  * rfc7748(): rfc7748.c needs modpro/modinv (addchain) so it is unbuildable here without a
    stand-in; the ladder fixtures are the reference's own KAT keys (rfc7748.c:271-277,
    simd/rfc7748_simt.cu:244-249) with the RFC 7748 answers, RFC 7748 5.2 iteration vectors,
    the reference main()'s LCG chain (rfc7748.c:297-305) and seeded random pairs, all computed
    with an independent big-integer model of RFC 7748 section 5 written below.  Output bytes
    are canonical, so this pins rfc7748() completely.

Inputs are drawn from random.Random(seed) so the run is reproducible.  Only data is written.
"""
import ctypes, json, os, random, sys
from ctypes import c_uint64, c_int, c_uint, c_char, POINTER

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refgen  # noqa: E402
import gio  # noqa: E402  (tests/golden/gio.py: plain or gzip-compressed fixture files)

U64P = POINTER(c_uint64)

HARNESS = r"""
/* harness (ours): the reference's time.c loop shapes around the reference-emitted functions */
unsigned int chain_modmul(spint *x, spint *y, long outer) {
    spint z[Nlimbs]; long i; int j;
    nres(x,x); nres(y,y);
    for (i=0;i<outer;i++) for (j=0;j<200;j++) {
        modmul(x,y,z); modmul(z,x,y); modmul(y,z,x); modmul(x,y,z); modmul(z,x,y);
    }
    redc(z,z);
    for (j=0;j<Nlimbs;j++) x[j]=z[j];
    return (unsigned int)(z[0]&0xFFFFFF);
}
unsigned int chain_modsqr(spint *x, long outer) {
    spint z[Nlimbs]; long i; int j;
    nres(x,x);
    for (i=0;i<outer;i++) for (j=0;j<500;j++) { modsqr(x,z); modsqr(z,x); }
    redc(z,z);
    for (j=0;j<Nlimbs;j++) x[j]=z[j];
    return (unsigned int)(z[0]&0xFFFFFF);
}
"""


def hx(v):
    return "0x%x" % v


class Ref:
    """ctypes view of the reference-emitted field code for one prime."""

    def __init__(self, script, prime, overrides=None):
        self.ns = refgen.load(script, 64, prime, overrides)
        self.lib, self.csrc = refgen.build(self.ns, HARNESS)
        ns = self.ns
        self.N, self.base, self.n, self.p = ns["N"], ns["base"], ns["n"], ns["p"]
        self.Nbytes = ns["Nbytes"]
        L = self.lib
        for f in ("modadd", "modsub", "modmul"):
            getattr(L, f).argtypes = [U64P, U64P, U64P]
        for f in ("modneg", "modsqr", "modcpy", "nres", "redc"):
            getattr(L, f).argtypes = [U64P, U64P]
        L.modmli.argtypes = [U64P, c_int, U64P]
        L.modnsqr.argtypes = [U64P, c_int]
        for f in ("flatten", "modfsb"):
            getattr(L, f).argtypes = [U64P]
            getattr(L, f).restype = c_uint64
        for f in ("modis1", "modis0", "modsign"):
            getattr(L, f).argtypes = [U64P]
            getattr(L, f).restype = c_int
        L.modcmp.argtypes = [U64P, U64P]
        L.modcmp.restype = c_int
        for f in ("modzer", "modone", "modhaf"):
            getattr(L, f).argtypes = [U64P]
        L.modint.argtypes = [c_int, U64P]
        L.mod2r.argtypes = [c_uint, U64P]
        L.modcmv.argtypes = [c_int, U64P, U64P]
        L.modcsw.argtypes = [c_int, U64P, U64P]
        L.modshl.argtypes = [c_uint, U64P]
        L.modshr.argtypes = [c_uint, U64P]
        L.modshr.restype = c_int
        L.modexp.argtypes = [U64P, POINTER(c_char)]
        L.modimp.argtypes = [POINTER(c_char), U64P]
        L.modimp.restype = c_int
        L.chain_modmul.argtypes = [U64P, U64P, ctypes.c_long]
        L.chain_modmul.restype = c_uint
        L.chain_modsqr.argtypes = [U64P, ctypes.c_long]
        L.chain_modsqr.restype = c_uint

    def arr(self, limbs=None):
        return (c_uint64 * self.N)(*(limbs or [0] * self.N))

    def split(self, x):
        """integer -> limbs, top limb unmasked (pseudo.py:1769-1775)."""
        b = 1 << self.base
        out = []
        for _ in range(self.N - 1):
            out.append(x % b)
            x >>= self.base
        out.append(x)
        return out

    def un(self, f, a):
        z = self.arr()
        getattr(self.lib, f)(self.arr(a), z)
        return list(z)

    def bi(self, f, a, b):
        z = self.arr()
        getattr(self.lib, f)(self.arr(a), self.arr(b), z)
        return list(z)

    def inplace(self, f, a, *pre):
        z = self.arr(a)
        r = getattr(self.lib, f)(*pre, z)
        return list(z), r


def params_of(ref):
    ns = ref.ns
    keys_common = ["WL", "n", "base", "N", "xcess", "Nbytes", "PM1D2", "PE", "p", "m"]
    keys_pm = ["mm", "TW", "EPM", "fred", "overflow", "carry_on", "bad_overflow_mul", "bad_overflow_sqr"]
    keys_mo = ["E", "R", "ndash", "trin", "M", "PM", "fullmonty"]
    out = {}
    for k in keys_common + keys_pm + keys_mo:
        if k in ns:
            v = ns[k]
            out[k] = hx(v) if isinstance(v, int) and not isinstance(v, bool) and v > 1 << 20 else v
    if "ppw" in ns:
        out["ppw"] = [("-" + hx(-v)) if v < 0 else hx(v) for v in ns["ppw"]]
        out["cw"] = [hx(v) for v in ns["cw"]]
    out["ROI"] = [hx(v) for v in ns["ROI"]]
    out["header"] = [l for l in ref.csrc.splitlines() if l.startswith("#define")][:12]
    out["log"] = ns["_log"].strip().splitlines()
    return out


def element_pool(ref, rng, count):
    """limb arrays covering the contract: canonical, [p,2p) unmasked-top, corners (edge.py:343-359 style)."""
    p, n, base = ref.p, ref.n, ref.base
    vals = []
    corners = [0, 1, 2, 3, p - 1, p - 2, p, p + 1, 2 * p - 1, 2 * p - 2, (p - 1) // 2, (p + 1) // 2,
               1 << 64, (1 << 64) - 1, 1 << (n - 1), (1 << (n - 1)) - 1, (1 << n) - 1 if (1 << n) - 1 < 2 * p else p - 3,
               (1 << n) - p, (1 << base) - 1, 1 << base, ((1 << base) - 1) << base]
    r = rng.randrange(0, p)
    corners += [r, pow(r, -1, p), (r + p)]
    vals += [c for c in corners if 0 <= c < 2 * p]
    while len(vals) < count:
        k = len(vals) % 3
        if k == 0:
            vals.append(rng.randrange(0, p))
        elif k == 1:
            vals.append(rng.randrange(0, 2 * p))
        else:
            vals.append(rng.randrange(p, 2 * p))
    return [ref.split(v) for v in vals[:count]], vals[:count]


def field_fixture(script, prime, seed, count=160, full_time=True, name=None):
    rng = random.Random(seed)
    ref = Ref(script, prime)
    prime = name or prime
    N, p = ref.N, ref.p
    fx = {"prime": prime, "generator": script, "generic": True, "params": params_of(ref), "seed": seed}
    araw, aval = element_pool(ref, rng, count)
    braw, bval = element_pool(ref, rng, count)
    rng.shuffle(braw)
    H = lambda limbs: [hx(v) for v in limbs]
    ops = {}
    # to internal form: nres of raw values (also a parity vector for nres itself on non-canonical input)
    A = [ref.un("nres", a) for a in araw]
    B = [ref.un("nres", b) for b in braw]
    # second half of the pool: use raw limbs directly as internal-form operands (top limb unmasked, < 2p)
    for i in range(count // 2, count):
        A[i] = araw[i]
        B[i] = braw[i]
    fx["raw_a"] = [H(a) for a in araw]
    fx["raw_b"] = [H(b) for b in braw]
    fx["nres_raw_a"] = [H(ref.un("nres", a)) for a in araw]
    fx["A"] = [H(a) for a in A]
    fx["B"] = [H(b) for b in B]
    for f in ("modadd", "modsub", "modmul"):
        ops[f] = [H(ref.bi(f, a, b)) for a, b in zip(A, B)]
    for f in ("modneg", "modsqr", "redc", "nres"):
        ops[f] = [H(ref.un(f, a)) for a in A]
    # chained, non-canonical: outputs fed back (what time.c and the ladder do)
    C = [ref.bi("modmul", a, b) for a, b in zip(A, B)]
    D = [ref.bi("modsub", a, b) for a, b in zip(A, B)]
    Ee = [ref.bi("modadd", a, b) for a, b in zip(A, B)]
    ops["chain_mul_CD"] = [H(ref.bi("modmul", c, d)) for c, d in zip(C, D)]
    ops["chain_sqr_D"] = [H(ref.un("modsqr", d)) for d in D]
    ops["chain_add_DE"] = [H(ref.bi("modadd", d, e)) for d, e in zip(D, Ee)]
    ops["chain_sub_EC"] = [H(ref.bi("modsub", e, c)) for e, c in zip(Ee, C)]
    ops["chain_redc_C"] = [H(ref.un("redc", c)) for c in C]
    # aliasing (legal and used: pseudo.py:752,1783,1793; rfc7748.c:214)
    al = []
    for a, b in zip(A[:16], B[:16]):
        x = ref.arr(a)
        y = ref.arr(b)
        ref.lib.modmul(x, y, x)
        ref.lib.modsqr(x, x)
        ref.lib.modadd(x, x, x)
        ref.lib.modsub(x, y, x)
        al.append(H(list(x)))
    ops["alias_chain"] = al
    # modmli by small ints (A24 of both curves, edge values); negative b sign-extends in the reference
    ints = [0, 1, 2, 19, 121665, 39081, 121666, 0x7fffffff, 65536, 3]
    ops["modmli_ints"] = ints
    ops["modmli"] = [[H(_mli(ref, a, k)) for k in ints] for a in A[:48]]
    # in-place predicates / normalisers
    ops["modfsb"] = [[H(z), int(r)] for z, r in (ref.inplace("modfsb", a) for a in A)]
    ops["flatten"] = [[H(z), int(r)] for z, r in (ref.inplace("flatten", a) for a in A)]
    ops["modis1"] = [int(ref.lib.modis1(ref.arr(a))) for a in A]
    ops["modis0"] = [int(ref.lib.modis0(ref.arr(a))) for a in A]
    ops["modsign"] = [int(ref.lib.modsign(ref.arr(a))) for a in A]
    ops["modcmp"] = [int(ref.lib.modcmp(ref.arr(a), ref.arr(b))) for a, b in zip(A, B)]
    ops["modcmp_self"] = [int(ref.lib.modcmp(ref.arr(a), ref.arr(ref.bi("modadd", a, ref.split(0))))) for a in A[:32]]
    ops["modhaf"] = [H(ref.inplace("modhaf", a)[0]) for a in A]
    sh = []
    for i, a in enumerate(A[:64]):
        k = 1 + i % 8
        zl, _ = ref.inplace("modshl", ref.un("redc", a), c_uint(k))
        zr, r = ref.inplace("modshr", a, c_uint(k))
        sh.append({"k": k, "shl_of_redc": H(zl), "shr": H(zr), "shr_ret": int(r)})
    ops["shifts"] = sh
    # conditional move / swap
    cs = []
    for i, (a, b) in enumerate(zip(A[:32], B[:32])):
        d = i & 1
        g, f = ref.arr(a), ref.arr(b)
        ref.lib.modcsw(d, g, f)
        f2 = ref.arr(b)
        ref.lib.modcmv(d, ref.arr(a), f2)
        cs.append({"d": d, "csw_g": H(list(g)), "csw_f": H(list(f)), "cmv_f": H(list(f2))})
    ops["cond"] = cs
    # constants
    z = ref.arr(); ref.lib.modone(z); ops["modone"] = H(list(z))
    z = ref.arr([7] * N); ref.lib.modzer(z); ops["modzer"] = H(list(z))
    ops["modint"] = [[k, H(ref.inplace("modint", [0] * N, c_int(k))[0])] for k in (0, 1, 2, 9, 5, 121665, 39081)]
    ops["mod2r"] = [[k, H(ref.inplace("mod2r", [0] * N, c_uint(k))[0])] for k in (0, 1, 51, 52, 56, 64, 100, 255, 256, 447, 448, 8 * ref.Nbytes - 1, 8 * ref.Nbytes)]
    # byte import / export (big-endian, pseudo.py:1115-1146)
    io_ = []
    for i in range(48):
        if i < 6:
            v = [0, 1, p - 1, p, p + 1, (1 << (8 * ref.Nbytes)) - 1][i]
        else:
            v = rng.randrange(0, 1 << (8 * ref.Nbytes)) if i % 2 else rng.randrange(0, p)
        bs = v.to_bytes(ref.Nbytes, "big")
        buf = (c_char * ref.Nbytes)(*bs)
        z = ref.arr()
        r = ref.lib.modimp(buf, z)
        out = (c_char * ref.Nbytes)()
        ref.lib.modexp(z, out)
        io_.append({"bytes": bs.hex(), "imp": H(list(z)), "imp_ret": int(r), "exp": bytes(out).hex()})
    ops["bytes"] = io_
    ops["modexp_A"] = []
    for a in A[:48]:
        out = (c_char * ref.Nbytes)()
        ref.lib.modexp(ref.arr(a), out)
        ops["modexp_A"].append(bytes(out).hex())
    # chain-independent pins for modinv / modsqrt / modqr (after redc)
    R = ref.ns.get("R", 1)
    Rinv = pow(R, -1, p)
    inv = []
    for a in A[:48]:
        val = _value(ref, a) * Rinv % p  # value represented by internal-form a
        iv = pow(val, -1, p) if val else 0
        qr = 1 if val == 0 or pow(val, (p - 1) // 2, p) == 1 else 0
        inv.append({"x": H(a), "inv_redc": H(_canon(ref, iv)), "qr": qr, "value": hx(val)})
    ops["modinv"] = inv
    fx["ops"] = ops
    # time.c protocol (pseudo.py:1862-1866 operands; check words pseudo.py:1250)
    r42 = random.Random()
    r42.seed(42)
    ra, rb, rs, ri = (r42.randint(0, p - 1) for _ in range(4))
    tm = {"ra": hx(ra), "rb": hx(rb), "rs": hx(rs), "ri": hx(ri)}
    b = 1 << ref.base
    mk = lambda v: [(v >> (ref.base * i)) % b for i in range(N)]  # makebig (pseudo.py:190-199)
    for outer, tag in ((1, "1k"), (100, "100k"), (100000, "full"))[:3 if full_time else 2]:
        x, y = ref.arr(mk(ra)), ref.arr(mk(rb))
        tm["modmul_check_" + tag] = hx(ref.lib.chain_modmul(x, y, outer))
        tm["modmul_z_" + tag] = H(list(x))
        x = ref.arr(mk(rs))
        tm["modsqr_check_" + tag] = hx(ref.lib.chain_modsqr(x, outer))
        tm["modsqr_z_" + tag] = H(list(x))
    tm["modinv_check_full"] = hx(pow(ri, -1, p) & 0xFFFFFF)  # z = 1/ri after an even number of inversions
    tm["modinv_z_full"] = H(_canon(ref, pow(ri, -1, p)))
    fx["time"] = tm
    return fx, ref


def _mli(ref, a, k):
    z = ref.arr()
    ref.lib.modmli(ref.arr(a), c_int(k), z)
    return list(z)


def _value(ref, limbs):
    return sum(v << (ref.base * i) for i, v in enumerate(limbs))


def _canon(ref, v):
    b = 1 << ref.base
    return [(v >> (ref.base * i)) % b for i in range(ref.N)]


def lazy_fixture(script, prime, seed, count=64):
    """generic=False variants (rfc7748.c:20; pseudo.py:294-325, 1523-1528): lazy modadd/modsub/modneg."""
    rng = random.Random(seed)
    ref = Ref(script, prime, overrides={"generic": False})
    assert ref.ns["algorithm"] is True and ref.ns["mp"] == 2
    H = lambda limbs: [hx(v) for v in limbs]
    A = [ref.split(rng.randrange(0, 2 * ref.p)) for _ in range(count)]
    B = [ref.split(rng.randrange(0, 2 * ref.p)) for _ in range(count)]
    fx = {"prime": prime, "generator": script, "generic": False, "mp": ref.ns["mp"], "A": [H(a) for a in A], "B": [H(b) for b in B], "ops": {}}
    for f in ("modadd", "modsub"):
        fx["ops"][f] = [H(ref.bi(f, a, b)) for a, b in zip(A, B)]
    fx["ops"]["modneg"] = [H(ref.un("modneg", a)) for a in A]
    return fx


# ---------------------------------------------------------------- RFC 7748 big-integer model
def _x_ladder(k_bytes, u_bytes, bits, p, a24, cof):
    """RFC 7748 section 5 (decodeScalar / decodeUCoordinate / ladder), independent of the reference."""
    nb = (bits + 7) // 8
    k = bytearray(k_bytes)
    k[0] &= 256 - (1 << cof)
    if bits % 8:
        k[nb - 1] &= (1 << (bits % 8)) - 1
        k[nb - 1] |= 1 << (bits % 8 - 1)
    else:
        k[nb - 1] |= 0x80
    kk = int.from_bytes(k, "little")
    ub = bytearray(u_bytes)
    if bits % 8:
        ub[nb - 1] &= (1 << (bits % 8)) - 1
    u = int.from_bytes(ub, "little") % p
    x1, x2, z2, x3, z3, swap = u, 1, 0, u, 1, 0
    for t in range(bits - 1, -1, -1):
        kt = (kk >> t) & 1
        swap ^= kt
        if swap:
            x2, x3, z2, z3 = x3, x2, z3, z2
        swap = kt
        A = (x2 + z2) % p; AA = A * A % p; B = (x2 - z2) % p; BB = B * B % p
        E = (AA - BB) % p; C = (x3 + z3) % p; D = (x3 - z3) % p
        DA = D * A % p; CB = C * B % p
        x3 = (DA + CB) ** 2 % p; z3 = x1 * (DA - CB) ** 2 % p
        x2 = AA * BB % p; z2 = E * (AA + a24 * E) % p
    if swap:
        x2, x3, z2, z3 = x3, x2, z3, z2
    return (x2 * pow(z2, p - 2, p) % p).to_bytes(nb, "little")


CURVES = {"X25519": (255, 2**255 - 19, 121665, 3, 9), "X448": (448, 2**448 - 2**224 - 1, 39081, 2, 5)}


def ladder(curve, k, u):
    bits, p, a24, cof, _ = CURVES[curve]
    return _x_ladder(k, u, bits, p, a24, cof)


def ladder_fixture(curve, seed, count=96):
    bits, p, a24, cof, gen = CURVES[curve]
    nb = (bits + 7) // 8
    rng = random.Random(seed)
    fx = {"curve": curve, "source": "RFC 7748 model (tests/golden/make_golden.py); reference KAT keys rfc7748.c:271-277"}
    G = bytes([gen]) + bytes(nb - 1)
    kat = []
    if curve == "X25519":
        sk = bytes.fromhex("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a")
        sk2 = bytes.fromhex("5dab087e624a8a4b79e17f8b83800ee66f3bb1292618b6fd1c2f8b27ff88e0eb")
        exp1 = "8520f0098930a754748b7ddcb43ef75a0dbf3a0d26381af4eba4a98eaa9b4e6a"
        exp2 = "de9edb7d7b7dc1b4d35b61c2ece435373f8343c85b78674dadfc7e146f882b4f"
        shared = "4a5d9d5ba4ce2de1728e3bf480350f25e07e21c947d19e3376f09b3c1e161742"
        it1 = "422c8e7a6227d7bca1350b3e2bb7279f7897b87bb6854b783c60e80311ae3079"
        it1000 = "684cf59ba83309552800ef566f2f4d3c1c3887c49360e3875f2eb94d99532c51"
        tv = [("a546e36bf0527c9d3b16154b82465edd62144c0ac1fc5a18506a2244ba449ac4",
               "e6db6867583030db3594c1a424b15f7c726624ec26b3353b10a903a6d0ab1c4c",
               "c3da55379de9c6908e94ea4df28d084f32eccf03491c71f754b4075577a28552"),
              ("4b66e9d4d1b4673c5ad22691957d6af5c11b6421e0ea01d42ca4169e7918ba0d",
               "e5210f12786811d3f4b7959d0538ae2c31dbe7106fc03c3efc4cd549c715a493",
               "95cbde9476e8907d7aade45cb4b873f88b595a68799fa152e6f8f7647aac7957")]
    else:
        sk = bytes.fromhex("9a8f4925d1519f5775cf46b04b5800d4ee9ee8bae8bc5565d498c28dd9c9baf574a9419744897391006382a6f127ab1d9ac2d8c0a598726b")
        sk2 = bytes.fromhex("1c306a7ac2a0e2e0990b294470cba339e6453772b075811d8fad0d1d6927c120bb5ee8972b0d3e21374c9c921b09d1b0366f10b65173992d")
        exp1 = "9b08f7cc31b7e3e67d22d5aea121074a273bd2b83de09c63faa73d2c22c5d9bbc836647241d953d40c5b12da88120d53177f80e532c41fa0"
        exp2 = "3eb7a829b0cd20f5bcfc0b599b6feccf6da4627107bdb0d4f345b43027d8b972fc3e34fb4232a13ca706dcb57aec3dae07bdc1c67bf33609"
        shared = "07fff4181ac6cc95ec1c16a94a0f74d12da232ce40a77552281d282bb60c0b56fd2464c335543936521c24403085d59a449a5037514a879d"
        it1 = "3f482c8a9f19b01e6c46ee9711d9dc14fd4bf67af30765c2ae2b846a4d23a8cd0db897086239492caf350b51f833868b9bc2b3bca9cf4113"
        it1000 = "aa3b4749d55b9daf1e5b00288826c467274ce3ebbdd5c17b975e09d4af6c67cf10d087202db88286e2b79fceea3ec353ef54faa26e219f38"
        tv = [("3d262fddf9ec8e88495266fea19a34d28882acef045104d0d1aae121700a779c984c24f8cdd78fbff44943eba368f54b29259a4f1c600ad3",
               "06fce640fa3487bfda5f6cf2d5263f8aad88334cbd07437f020f08f9814dc031ddbdc38c19c6da2583fa5429db94ada18aa7a7fb4ef8a086",
               "ce3e4ff95a60dc6697da1db1d85e6afbdf79b50a2412d7546d5f239fe14fbaadeb445fc66a01b0779d98223961111e21766282f73dd96b6f"),
              ("203d494428b8399352665ddca42f9de8fef600908e0d461cb021f8c538345dd77c3e4806e25f46d3315c44e0a5b4371282dd2c8d5be3095f",
               "0fbcc2f993cd56d3305b0b7d9e55d4c1a8fb5dbb52f8e9a1e9b6201b165d015894e56c4d3570bee52fe205e28a78b91cdfbde71ce8d157db",
               "884a02576239ff7a2f2f63b2db6a9ff37047ac13568e1e30fe63c4a7ad1b3ee3a5700df34321d62077e63633c575c1c954514e99da7c179d")]
    assert ladder(curve, sk, G).hex() == exp1
    assert ladder(curve, sk2, G).hex() == exp2
    assert ladder(curve, sk, bytes.fromhex(exp2)).hex() == shared
    kat.append({"k": sk.hex(), "u": G.hex(), "out": exp1, "src": "rfc7748.c:271-277 key; RFC 7748 6.x public key"})
    kat.append({"k": sk2.hex(), "u": G.hex(), "out": exp2, "src": "simd/rfc7748_simt.cu:245 key (X25519); RFC 7748 6.x"})
    kat.append({"k": sk.hex(), "u": exp2, "out": shared, "src": "RFC 7748 6.x shared secret"})
    for k_, u_, o_ in tv:
        assert ladder(curve, bytes.fromhex(k_), bytes.fromhex(u_)).hex() == o_
        kat.append({"k": k_, "u": u_, "out": o_, "src": "RFC 7748 5.2 test vector"})
    # RFC 7748 5.2 iteration test (1 and 1000 iterations)
    k = u = G
    for i in range(1000):
        k, u = ladder(curve, k, u), k
        if i == 0:
            assert k.hex() == it1
    assert k.hex() == it1000
    fx["iter"] = {"start": G.hex(), "after_1": it1, "after_1000": it1000}
    # the reference main()'s own chain: LCG key, 5000 x (bk,bu->bv ; bk,bv->bu)  (rfc7748.c:297-305)
    rnd = 1
    bk = bytearray(nb)
    for i in range(nb):
        rnd = (5 * rnd + 1) & 0xFFFF
        bk[i] = rnd % 256
    bu = G
    chain = {"bk": bytes(bk).hex(), "bu0": G.hex(), "checkpoints": {}}
    for i in range(5000):
        bv = ladder(curve, bytes(bk), bu)
        bu = ladder(curve, bytes(bk), bv)
        if i + 1 in (1, 10, 100, 1000, 5000):
            chain["checkpoints"][str(i + 1)] = bu.hex()
    # DH exchange that follows (rfc7748.c:321-333)
    alice, bob = bytearray(nb), bytearray(nb)
    for i in range(nb):
        rnd = (5 * rnd + 1) & 0xFFFF; alice[i] = rnd % 256
        rnd = (5 * rnd + 1) & 0xFFFF; bob[i] = rnd % 256
    apk = ladder(curve, bytes(alice), G); bpk = ladder(curve, bytes(bob), G)
    ssa = ladder(curve, bytes(alice), bpk); ssb = ladder(curve, bytes(bob), apk)
    assert ssa == ssb
    chain["dh"] = {"alice": bytes(alice).hex(), "bob": bytes(bob).hex(), "apk": apk.hex(), "bpk": bpk.hex(), "shared": ssa.hex()}
    fx["ref_main_chain"] = chain
    # seeded random pairs incl. awkward u: zero, one, p-1, p, p+1, 2^bits-1 (non-canonical), high bit set
    pairs = []
    special_u = [0, 1, p - 1, p, p + 1, (1 << (8 * nb)) - 1, (1 << bits) - 1, 2, gen]
    for i in range(count):
        kb = bytes(rng.randrange(256) for _ in range(nb))
        if i < len(special_u):
            ub = (special_u[i] % (1 << (8 * nb))).to_bytes(nb, "little")
        else:
            ub = bytes(rng.randrange(256) for _ in range(nb))
        pairs.append({"k": kb.hex(), "u": ub.hex(), "out": ladder(curve, kb, ub).hex()})
    pairs.append({"k": bytes(nb).hex(), "u": G.hex(), "out": ladder(curve, bytes(nb), G).hex()})
    pairs.append({"k": (b"\xff" * nb).hex(), "u": G.hex(), "out": ladder(curve, b"\xff" * nb, G).hex()})
    fx["kat"] = kat
    fx["pairs"] = pairs
    return fx


def sqrt_model(val, p, pm1d2, pe, roi):
    """value-level restatement of the generators' modsqrt (pseudo.py:834-874): deterministic for
    residues and non-residues alike; independent of the addition chain."""
    y = pow(val, pe, p)
    s = y * val % p
    if pm1d2 > 1:
        t = s * y % p
        z = roi
        for k in range(pm1d2, 1, -1):
            b = pow(t, 1 << (k - 2), p)
            d = 0 if b == 1 else 1
            if d:
                s = s * z % p
            z = z * z % p
            if d:
                t = t * z % p
    return s


def sqrt_fixture(script, prime, seed, count=64, name=None):
    """modsqrt / modqr pins (the reference cannot emit them here: they call modpro, which needs the
    external addchain tool).  Inputs are internal-form limbs produced by the reference's own nres;
    expectations are big-integer values: root after redc, Euler criterion."""
    rng = random.Random(seed)
    ref = Ref(script, prime)
    p, ns = ref.p, ref.ns
    roi = sum(v << (ref.base * i) for i, v in enumerate(ns["ROI"]))
    H = lambda limbs: [hx(v) for v in limbs]
    recs = []
    vals = [0, 1, 4, p - 1, 2, 3] + [rng.randrange(0, p) for _ in range(count - 6)]
    for i, v in enumerate(vals):
        if i % 3 == 2:
            v = v * v % p                                  # force residues into the mix
        x = ref.un("nres", ref.split(v + (p if i % 5 == 4 else 0)))   # some inputs in [p, 2p)
        root = sqrt_model(v, p, ns["PM1D2"], ns["PE"], roi)
        qr = 1 if v == 0 or pow(v, (p - 1) // 2, p) == 1 else 0
        if qr:
            assert root * root % p == v
        recs.append({"x": H(x), "value": hx(v), "sqrt_redc": H(_canon(ref, root)), "qr": qr})
    return {"prime": name or prime, "source": "value-level model of pseudo.py:815-874 on reference-nres'd inputs", "recs": recs}


# ---------------------------------------------------------------- Edwards big-integer model
class EdModel:
    """a*x^2 + y^2 = 1 + d*x^2*y^2 over GF(p), extended coordinates on python ints.  Independent of the
    reference code; constants are the ones curve.py:85-105 lists."""

    def __init__(self, name):
        if name == "ED25519":
            self.p = 2**255 - 19
            self.a, self.cof = -1, 3
            self.d = 0x52036CEE2B6FFE738CC740797779E89800700A4D4141D8AB75EB4DCA135978A3
            self.q = 0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED
            self.G = (0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A,
                      0x6666666666666666666666666666666666666666666666666666666666666658)
            self.nbytes = 32
            # testcurve.c:57-63
            self.tc = dict(order="1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED",
                           r1="66876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                           r2="9978934937938999F9998765C8909870B885907FDF03764C13B05B94EE93672",
                           n1="20347457078878f77b707c070707077a07707b7b07070707223252357134272",
                           n2="35279279432f249b298a876788d86294e02842092769136c086038b1812383a")
        elif name == "NUMS256E":                      # curve.py:137-145: generator from x = 34, y of even sign
            self.p = 2**256 - 189
            self.a, self.cof = 1, 2
            self.d = -15342 % self.p
            self.q = 0x4000000000000000000000000000000041955AA52F59439B1A47B190EEDD4AF5
            self.nbytes = 32
            self.G = (34, self.recover_y(34, 0))
            # testcurve.c:50-56
            self.tc = dict(order="4000000000000000000000000000000041955AA52F59439B1A47B190EEDD4AF5",
                           r1="166876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                           r2="29978934937938999F9998765C890987383EB9CE8A51DE298370542FE0D0AD7A",
                           n1="21347457078878f77b707c070707077a07707b7b070707072232523571342729",
                           n2="35279279432f249b298a876788d86294e02842092769136c086038b1812383a5")
        elif name in ("ED248", "ED376", "ED500"):     # curve.py:107-135: generator from a small x, y of even sign
            self.p, d, gx, self.q, self.nbytes, self.tc = {
                # testcurve.c:85-105
                "ED248": (5 * 2**248 - 1, -107431, 4, 0x13FFFFFFFFFFFFFFFFFFFFFFFFFFFFFF098677E8D0D856DA332BA970DCFDEA1, 32,
                          dict(order="13FFFFFFFFFFFFFFFFFFFFFFFFFFFFFF098677E8D0D856DA332BA970DCFDEA1",
                               r1="10876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                               r2="378934937938999f9998765c890986e741c6a7e8061ffc0c5b5d35ffc34126",
                               n1="7347457078878f77b707c070707077a07707b7b07070707223252357134272",
                               n2="3279279432f249b298a876788d86294e02842092769136c086038b1812383a")),
                "ED376": (65 * 2**376 - 1, -66524, 2,
                          0x104000000000000000000000000000000000000000000000303A69B3514879CD109A98F29F0D04F09F855D4F3C6A7037, 48,
                          dict(order="104000000000000000000000000000000000000000000000303A69B3514879CD109A98F29F0D04F09F855D4F3C6A7037",
                               r1="6076CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B87508750F98765C890986DAE5E19F451E",
                               r2="a38934937938999f9998765c890986f6a95f295af89a8e6c2c493a2707ea2149b9223e30696a86795fe82695acb2b19",
                               n1="93347457078878f77b707c070707077a07707b7b070707072232523571342720986DAE5E19F451EF6EE89D3C2C0986D",
                               n2="235279279432f249b298a876788d86294e02842092769136c086038b1812383a13427294582948924358f98985956A0")),
                "ED500": (27 * 2**500 - 1, -105355, 6,
                          0x6C00000000000000000000000000000000000000000000000000000000000002C8858DA0CB07C5ABCADABC1BEE86F8C9101174D8A115AD57E5F0228C2D0871, 64,
                          dict(order="6C00000000000000000000000000000000000000000000000000000000000002C8858DA0CB07C5ABCADABC1BEE86F8C9101174D8A115AD57E5F0228C2D0871",
                               r1="47235279279432f249b298a876788d86294e02842092769136c086038b1812383a8765C890985B1583C100A413ACA28FB0654735836726356262066876CB6C",
                               r2="24dcad86d86bcd0db64d675789877279d6b1fd7bdf6d896ec93f79fc74e7edca8dfe27d83a6f6a964719bb77dada56395fac2da31dae8722838e1c23b63d05",
                               n1="32120347457078878f77b707c070707077a07707b7b0707070722325235713427270707077a07707b7b07070707223252307707b7b070707072232523688A3",
                               n2="19235279279432f249b298a876788d86294e02842092769136c086038b1812383a13427294582948924358f98985956A077b707c070707077a07707b7b0707")),
            }[name]
            self.a, self.cof = 1, 2
            self.d = d % self.p
            self.G = (gx, self.recover_y(gx, 0))
        else:
            self.p = 2**448 - 2**224 - 1
            self.a, self.cof = 1, 2
            self.d = -39081 % self.p
            self.q = (self.p + 1 - 28312320572429821613362531907042076847709625476988141958474579766324) // 4
            self.G = (0x4f1970c66bed0ded221d15a622bf36da9e146570470f1767ea6de324a3d3a46412ae1af72ab66511433b80e18b00938e2626a82bc70cc05e,
                      0x693f46716eb6bc248876203756c9c7624bea73736ca3984087789c1e05a0c2d73ad3ff1ce67c39c4fdbd132c4ed7c8ad9808795bf230fa14)
            self.nbytes = 56
            # testcurve.c:64-70
            self.tc = dict(order="3FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF7CCA23E9C44EDB49AED63690216CC2728DC58F552378C292AB5844F3",
                           r1="1F6868465567867838578786787978654735836726356262066876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                           r2="209797B9AA987987C7A878798786879AB8CA7C98D9CA9D9DF997893410435C8363E873C00B5F40171816219BE8BE29E38CA165319D4BA778",
                           n1="21347457078878f77b707c070707077a07707b7b0707070722325235713427294582948924358f9898529852985298085690868659860866",
                           n2="35279279432f249b298a876788d86294e02842092769136c086038b1812383a5875793576874589355835357878467862937894039158588")
        assert self.on_curve(self.G)

    def on_curve(self, P):
        x, y = P
        return (self.a * x * x + y * y - 1 - self.d * x * x * y * y) % self.p == 0

    def add(self, P, Q):  # affine in, affine out (complete formulas)
        p, a, d = self.p, self.a, self.d
        x1, y1 = P
        x2, y2 = Q
        k = d * x1 * x2 * y1 * y2 % p
        x3 = (x1 * y2 + y1 * x2) * pow(1 + k, -1, p) % p
        y3 = (y1 * y2 - a * x1 * x2) * pow(1 - k, -1, p) % p
        return (x3, y3)

    # extended coordinates (X:Y:Z:T) for speed
    def _ext_add(self, P, Q):
        p, a, d = self.p, self.a, self.d
        X1, Y1, Z1, T1 = P
        X2, Y2, Z2, T2 = Q
        A = X1 * X2 % p; B = Y1 * Y2 % p; C = T1 * d * T2 % p; D = Z1 * Z2 % p
        E = ((X1 + Y1) * (X2 + Y2) - A - B) % p
        F = (D - C) % p; G = (D + C) % p; H = (B - a * A) % p
        return (E * F % p, G * H % p, F * G % p, E * H % p)

    def mul(self, k, P):
        O = (0, 1, 1, 0)
        Q = (P[0], P[1], 1, P[0] * P[1] % self.p)
        R = O
        for bit in bin(k)[2:] if k else "":
            R = self._ext_add(R, R)
            if bit == "1":
                R = self._ext_add(R, Q)
        zi = pow(R[2], -1, self.p)
        return (R[0] * zi % self.p, R[1] * zi % self.p)

    def neg(self, P):
        return ((-P[0]) % self.p, P[1])

    def xy_hex(self, P):
        return [P[0].to_bytes(self.nbytes, "big").hex(), P[1].to_bytes(self.nbytes, "big").hex()]

    def recover_y(self, x, s):
        """y from x and the parity s of y, or None if x is not on the curve"""
        p = self.p
        num = (1 - self.a * x * x) % p
        den = (1 - self.d * x * x) % p
        if den == 0:
            return None
        y2 = num * pow(den, -1, p) % p
        return self._sqrt_sign(y2, s)

    def recover_x(self, y, s):
        p = self.p
        num = (1 - y * y) % p
        den = (self.a - self.d * y * y) % p
        if den == 0:
            return None
        x2 = num * pow(den, -1, p) % p
        return self._sqrt_sign(x2, s)

    def _sqrt_sign(self, v, s):
        p = self.p
        if v == 0:
            return 0
        if pow(v, (p - 1) // 2, p) != 1:
            return None
        if p % 4 == 3:
            r = pow(v, (p + 1) // 4, p)
        else:
            r = pow(v, (p + 3) // 8, p)
            if r * r % p != v:
                r = r * pow(2, (p - 1) // 4, p) % p
        assert r * r % p == v
        if r % 2 != s:
            r = p - r
        return r


def edwards_fixture(name, seed, pairs=40):
    import hashlib
    rng = random.Random(seed)
    M = EdModel(name)
    nb, p, q, G = M.nbytes, M.p, M.q, M.G
    O = (0, 1)
    fx = {"curve": name, "source": "big-integer model (tests/golden/make_golden.py EdModel); constants curve.py:85-105",
          "gen": M.xy_hex(G), "order": q.to_bytes(nb, "big").hex(), "cof": M.cof}
    assert M.mul(q, G) == O
    # scalar multiplications of the generator and of random points; scalars are raw NBYTES strings (the
    # reference's fixed-window ecnXXXmul does not reduce them, edwards.c:435-482)
    recs = []
    for i in range(pairs):
        k = [0, 1, 2, 8, 15, 16, 17, q - 1, q, q + 1, (1 << (8 * nb)) - 1, 0x88888888, 0x77777777][i] if i < 13 else rng.randrange(0, 1 << (8 * nb))
        base = G if i % 2 == 0 else M.mul(rng.randrange(1, q), G)
        recs.append({"e": k.to_bytes(nb, "big").hex(), "P": M.xy_hex(base), "eP": M.xy_hex(M.mul(k, base))})
    fx["mul"] = recs
    # add / dbl / sub / neg
    ops = []
    for i in range(16):
        P = M.mul(rng.randrange(1, q), G)
        Q = M.mul(rng.randrange(1, q), G) if i % 4 else (P if i % 8 == 0 else M.neg(P))
        ops.append({"P": M.xy_hex(P), "Q": M.xy_hex(Q), "P+Q": M.xy_hex(M.add(P, Q)), "2P": M.xy_hex(M.add(P, P)),
                    "P-Q": M.xy_hex(M.add(P, M.neg(Q))), "cofP": M.xy_hex(M.mul(1 << M.cof, P))})
    fx["ops"] = ops
    # compression / decompression: x + sign(y) and y + sign(x); off-curve coordinates must give O
    comp = []
    for i in range(24):
        if i < 16:
            P = M.mul(rng.randrange(1, q), G)
            comp.append({"x": P[0].to_bytes(nb, "big").hex(), "y": P[1].to_bytes(nb, "big").hex(), "sx": P[0] & 1, "sy": P[1] & 1, "valid": 1})
        else:
            while True:
                v = rng.randrange(2, p)
                if M.recover_y(v, 0) is None and M.recover_x(v, 0) is None:
                    break
            comp.append({"x": v.to_bytes(nb, "big").hex(), "y": v.to_bytes(nb, "big").hex(), "sx": 0, "sy": 0, "valid": 0})
    fx["compress"] = comp
    # full (x,y) set with an off-curve pair
    bad = (G[0], (G[1] + 1) % p)
    fx["set_xy"] = [{"x": M.xy_hex(G)[0], "y": M.xy_hex(G)[1], "valid": 1}, {"x": M.xy_hex(bad)[0], "y": M.xy_hex(bad)[1], "valid": 0}]
    # mul2: R = eP + fQ
    m2 = []
    for i in range(12):
        P = M.mul(rng.randrange(1, q), G); Q = M.mul(rng.randrange(1, q), G)
        e = rng.randrange(0, 1 << (8 * nb)) if i else 0
        f = rng.randrange(0, 1 << (8 * nb)) if i != 1 else 0
        m2.append({"e": e.to_bytes(nb, "big").hex(), "f": f.to_bytes(nb, "big").hex(), "P": M.xy_hex(P), "Q": M.xy_hex(Q),
                   "R": M.xy_hex(M.add(M.mul(e, P), M.mul(f, Q)))})
    fx["mul2"] = m2
    if M.tc:
        t = {k: int(v, 16) for k, v in M.tc.items()}
        assert t["order"] == q and (t["r1"] + t["r2"]) == q
        tc = {k: v.to_bytes(nb, "big").hex() for k, v in t.items()}
        assert M.add(M.mul(t["r1"], G), M.mul(t["r2"], G)) == O
        # the reference main()'s timing chain (testcurve.c:247-255): P = G; 10000 x P = n1*P
        P = G
        cps = {}
        for i in range(10000):
            P = M.mul(t["n1"], P)
            if i + 1 in (1, 10, 100, 1000, 10000):
                cps[str(i + 1)] = M.xy_hex(P)
        tc["mul_chain"] = cps
        # then 10000 x P = n1*P + n2*G (testcurve.c:274-276); checkpoints only up to 100
        Q = G
        cps2 = {}
        for i in range(100):
            P = M.add(M.mul(t["n1"], P), M.mul(t["n2"], Q))
            if i + 1 in (1, 10, 100):
                cps2[str(i + 1)] = M.xy_hex(P)
        tc["mul2_chain_after_mul_chain"] = cps2
        fx["testcurve"] = tc
    if name == "ED25519":
        # RFC 8032 7.1 TEST 1: public key = compress([clamp(SHA512(sk)[:32])] B)
        sk = bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")
        h = bytearray(hashlib.sha512(sk).digest()[:32])
        h[0] &= 248; h[31] &= 127; h[31] |= 64
        a = int.from_bytes(h, "little")
        A = M.mul(a, G)
        enc = bytearray(A[1].to_bytes(32, "little"))
        enc[31] |= (A[0] & 1) << 7
        assert enc.hex() == "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"
        fx["rfc8032_test1"] = {"scalar_be": a.to_bytes(32, "big").hex(), "pk": enc.hex(), "A": M.xy_hex(A)}
    return fx


class WsModel:
    """y^2 = x^3 + a*x + b over GF(p), affine big-integer arithmetic with None as the point at infinity.
    Constants of curve.py:157-166 (NIST256)."""

    def __init__(self, name):
        if name == "NIST256":
            self.p = 2**256 - 2**224 + 2**192 + 2**96 - 1
            self.a = -3
            self.b = 0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b
            self.q = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
            self.G = (0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296,
                      0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5)
            self.nbytes = 32
            # testcurve.c:29-35
            self.tc = dict(order="FFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551",
                           r1="166876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                           r2="E99789339379389A9F9998765C890986B39059D7021039135CE26D61EE5687D6",
                           n1="C20347457078878f77b707c070707077a07707b7b07070707223252357134272",
                           n2="D35279279432f249b298a876788d86294e02842092769136c086038b1812383a")
        elif name == "NIST384":                       # curve.py:168-177
            self.p = 2**384 - 2**128 - 2**96 + 2**32 - 1
            self.a = -3
            self.b = 27580193559959705877849011840389048093056905856361568521428707301988689241309860865136260764883745107765439761230575
            self.q = self.p + 1 - 1388124618062372383606759648309780106643088307173319169677
            self.G = (0xaa87ca22be8b05378eb1c71ef320ad746e1d3b628ba79b9859f741e082542a385502f25dbf55296c3a545e3872760ab7,
                      0x3617de4a96262c6f5d9e98bf9292dc29f8f41dbd289a147ce9da3113b5f0b8c00a60b1ce1d7e819d7a431d7c90ea0e5f)
            self.nbytes = 48
            # testcurve.c:36-42
            self.tc = dict(order="ffffffffffffffffffffffffffffffffffffffffffffffffc7634d81f4372ddf581a0db248b0a77aecec196accc52973",
                           r1="bd9c66b3ad3c2d6d1a3d1fa7bc8960a923b8c1e9392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d",
                           r2="4263994c52c3d292e5c2e05843769f56dc473e16c6dba92188b211f1adcedb879a43ccb742498ca9d06be7eb2913afd6",
                           n1="9a1de644815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199972a846916419f828b9d2434e465e150",
                           n2="4737819096da1dac72ff5d2a386ecbe06b65a6a48b8148f6b38a088ca65ed389b74d0fb132e706298fadc1a606cb0fb3")
        elif name == "NIST521":                       # curve.py:179-188
            self.p = 2**521 - 1
            self.a = -3
            self.b = 0x51953EB9618E1C9A1F929A21A0B68540EEA2DA725B99B315F3B8B489918EF109E156193951EC7E937B1652C0BD3BB1BF073573DF883D2C34F1EF451FD46B503F00
            self.q = 0x1fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffa51868783bf2f966b7fcc0148f709a5d03bb5c9b8899c47aebb6fb71e91386409
            self.G = (0xC6858E06B70404E9CD9E3ECB662395B4429C648139053FB521F828AF606B4D3DBAA14B5E77EFE75928FE1DC127A2FFA8DE3348B3C1856A429BF97E7E31C2E5BD66,
                      0x11839296A789A3BC0045C8A5FB42C7D1BD998F54449579B446817AFBD17273E662C97EE72995EF42640C550B9013FAD0761353C7086A272C24088BE94769FD16650)
            self.nbytes = 66
            # testcurve.c:43-49
            self.tc = dict(order="1fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffa51868783bf2f966b7fcc0148f709a5d03bb5c9b8899c47aebb6fb71e91386409",
                           r1="d8972a846916419f828b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a923b8c1e9392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d",
                           r2="12768d57b96e9be607d7462dbcb1b9a1eaf4263994c52c3d292e5c2e05843769f512dcdc59a860b3f8d411ac5b8b0a153787ddf88bd83352cdd9eef859eed86ea6c",
                           n1="e5386ecbe06b65a6a48b8148f6b38a088ca65ed389b74d0fb132e706298fadc1a606cb0fb39a1de644815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f3",
                           n2="acc37459eef50bea63371ecd7b27cd813047229389571aa8766c307511b2b9437a28df6ec4ce4a2bbdc241330b01a9e71fde8a774bcf36d58b4737819096da1dac")
        elif name == "SECP256K1":                     # curve.py:190-198
            self.p = 2**256 - 2**32 - 977
            self.a = 0
            self.b = 7
            self.q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
            self.G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
                      0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
            self.nbytes = 32
            # testcurve.c:78-84
            self.tc = dict(order="FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141",
                           r1="166876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                           r2="E9978934937938999F9998765C890985B1583C100A413ACA28FB012BC229A3C6",
                           n1="120347457078878f77b707c070707077a07707b7b07070707223252357134272",
                           n2="235279279432f249b298a876788d86294e02842092769136c086038b1812383a")
        elif name == "NUMS256W":                      # curve.py:147-155: generator from x = 2, y of even sign
            self.p = 2**256 - 189
            self.a = -3
            self.b = 152961
            self.q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE43C8275EA265C6020AB20294751A825
            self.nbytes = 32
            self.G = (2, self.recover_y(2, 0))
            # testcurve.c:71-77
            self.tc = dict(order="FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE43C8275EA265C6020AB20294751A825",
                           r1="166876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B",
                           r2="E9978934937938999F9998765C890986DAE5E19F451EF6EE89D3C2C839450AAA",
                           n1="120347457078878f77b707c070707077a07707b7b07070707223252357134272",
                           n2="235279279432f249b298a876788d86294e02842092769136c086038b1812383a")
        else:
            raise ValueError(name)
        assert self.on_curve(self.G)

    def on_curve(self, P):
        x, y = P
        return (y * y - x * x * x - self.a * x - self.b) % self.p == 0

    def add(self, P, Q):
        p = self.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1 + self.a) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def neg(self, P):
        return None if P is None else (P[0], (-P[1]) % self.p)

    def mul(self, k, P):
        # Jacobian-free: double-and-add on affine points with python ints is fast enough for fixtures
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.add(R, R)
            if bit == "1":
                R = self.add(R, P)
        return R

    def xy_hex(self, P):
        """affine coordinates as the reference's ecnXXXget leaves them; infinity is (0, 1) after affine()"""
        if P is None:
            return [(0).to_bytes(self.nbytes, "big").hex(), (1).to_bytes(self.nbytes, "big").hex()]
        return [P[0].to_bytes(self.nbytes, "big").hex(), P[1].to_bytes(self.nbytes, "big").hex()]

    def recover_y(self, x, s):
        p = self.p
        v = (x * x * x + self.a * x + self.b) % p
        if v == 0:
            return 0
        if pow(v, (p - 1) // 2, p) != 1:
            return None
        r = pow(v, (p + 1) // 4, p)
        return r if r % 2 == s else p - r


def weierstrass_fixture(name, seed, pairs=32):
    rng = random.Random(seed)
    M = WsModel(name)
    nb, p, q, G = M.nbytes, M.p, M.q, M.G
    fx = {"curve": name, "source": "big-integer model (tests/golden/make_golden.py WsModel); constants curve.py:147-198",
          "gen": M.xy_hex(G), "order": q.to_bytes(nb, "big").hex()}
    assert M.mul(q, G) is None
    recs = []
    for i in range(pairs):
        k = [0, 1, 2, 8, 15, 16, 17, q - 1, q, q + 1, (1 << (8 * nb)) - 1, 0x88888888, 0x77777777][i] if i < 13 else rng.randrange(0, 1 << (8 * nb))
        base = G if i % 2 == 0 else M.mul(rng.randrange(1, q), G)
        R = M.mul(k, base)
        recs.append({"e": k.to_bytes(nb, "big").hex(), "P": M.xy_hex(base), "eP": M.xy_hex(R), "inf": int(R is None)})
    fx["mul"] = recs
    ops = []
    for i in range(16):
        P = M.mul(rng.randrange(1, q), G)
        Q = M.mul(rng.randrange(1, q), G) if i % 4 else (P if i % 8 == 0 else M.neg(P))
        S, D = M.add(P, Q), M.add(P, M.neg(Q))
        ops.append({"P": M.xy_hex(P), "Q": M.xy_hex(Q), "P+Q": M.xy_hex(S), "P+Q_inf": int(S is None), "2P": M.xy_hex(M.add(P, P)),
                    "P-Q": M.xy_hex(D), "P-Q_inf": int(D is None)})
    fx["ops"] = ops
    comp = []
    for i in range(20):
        if i < 14:
            P = M.mul(rng.randrange(1, q), G)
            comp.append({"x": P[0].to_bytes(nb, "big").hex(), "y": P[1].to_bytes(nb, "big").hex(), "sy": P[1] & 1, "valid": 1})
        else:
            while True:
                v = rng.randrange(2, p)
                if M.recover_y(v, 0) is None:
                    break
            comp.append({"x": v.to_bytes(nb, "big").hex(), "y": v.to_bytes(nb, "big").hex(), "sy": 0, "valid": 0})
    fx["compress"] = comp
    bad = (G[0], (G[1] + 1) % p)
    fx["set_xy"] = [{"x": M.xy_hex(G)[0], "y": M.xy_hex(G)[1], "valid": 1}, {"x": M.xy_hex(bad)[0], "y": M.xy_hex(bad)[1], "valid": 0}]
    m2 = []
    for i in range(10):
        P = M.mul(rng.randrange(1, q), G); Q = M.mul(rng.randrange(1, q), G)
        e = rng.randrange(0, 1 << (8 * nb)) if i else 0
        f = rng.randrange(0, 1 << (8 * nb)) if i != 1 else 0
        R = M.add(M.mul(e, P), M.mul(f, Q))
        m2.append({"e": e.to_bytes(nb, "big").hex(), "f": f.to_bytes(nb, "big").hex(), "P": M.xy_hex(P), "Q": M.xy_hex(Q),
                   "R": M.xy_hex(R), "inf": int(R is None)})
    fx["mul2"] = m2
    t = {k: int(v, 16) for k, v in M.tc.items()}
    assert t["order"] == q and (t["r1"] + t["r2"]) == q
    tc = {k: v.to_bytes(nb, "big").hex() for k, v in t.items()}
    assert M.add(M.mul(t["r1"], G), M.mul(t["r2"], G)) is None
    P = G
    cps = {}
    for i in range(1000):
        P = M.mul(t["n1"], P)
        if i + 1 in (1, 10, 100, 1000):
            cps[str(i + 1)] = M.xy_hex(P)
    tc["mul_chain"] = cps
    fx["testcurve"] = tc
    return fx


def extras():
    """further primes and group orders built by the engine (modarith_amd.emit.EXTRA_PRIMES): the same
    fixture recipe, fewer pairs, straight from the reference generators"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from modarith_amd.emit import EXTRA_PRIMES
    from modarith_amd.params import reference_argv
    only = [a for a in sys.argv[1:] if not a.startswith("-")]       # optional: regenerate just these
    for k, name in enumerate(EXTRA_PRIMES):
        if only and name not in only:
            continue
        script, arg = reference_argv(name)
        fx, _ = field_fixture(script, arg, 6000 + k, count=64, full_time=False, name=name)
        gio.dump(fx, "field_%s.json" % name)
        gio.dump(sqrt_fixture(script, arg, 7000 + k, count=24, name=name), "sqrt_%s.json" % name)
        print(name, fx["params"]["log"][0])


def generated():
    """the unnamed moduli of modarith_amd.generate.EXAMPLES, through the same recipe: what the reference's generators emit when
    they are handed an expression instead of a name (pseudo.py:1552-1556, monty.py:2113-2121)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from modarith_amd.generate import EXAMPLES, resolve
    for k, (arg, fam) in enumerate(EXAMPLES):
        tag = resolve(arg, fam).name
        expr = arg.split("=", 1)[-1]
        script = "pseudo.py" if fam == "pseudo" else "monty.py"
        fx, _ = field_fixture(script, expr, 9000 + k, count=64, full_time=False, name=tag)
        gio.dump(fx, "field_%s.json" % tag)
        gio.dump(sqrt_fixture(script, expr, 9100 + k, count=24, name=tag), "sqrt_%s.json" % tag)
        print(tag, fx["params"]["log"][0])


def main():
    if "--generated-only" in sys.argv:
        generated()
        return
    if "--extras-only" in sys.argv:
        extras()
        return
    if "--weierstrass-only" in sys.argv:
        for k, wname in enumerate(("NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W")):
            gio.dump(weierstrass_fixture(wname, 8001 + k, pairs=32 if k == 0 else (24, 20, 28, 28)[k - 1]), "weierstrass_%s.json" % wname)
        return
    if "--edwards-only" in sys.argv:
        for name, seed in (("ED25519", 5001), ("ED448", 5002), ("NUMS256E", 5003), ("ED248", 5004), ("ED376", 5005), ("ED500", 5006)):
            fx = edwards_fixture(name, seed, pairs={"ED25519": 40, "ED376": 16, "ED500": 12}.get(name, 20))
            gio.dump(fx, "edwards_%s.json" % name)
            if "testcurve" in fx:
                print(name, "testcurve chain 10000:", fx["testcurve"]["mul_chain"]["10000"])
        return
    if "--sqrt-only" in sys.argv:
        for script, prime, seed in (("pseudo.py", "X25519", 4001), ("monty.py", "NIST256", 4002), ("monty.py", "X448", 4003)):
            gio.dump(sqrt_fixture(script, prime, seed), "sqrt_%s.json" % prime)
        return
    out = {}
    for script, prime, seed in (("pseudo.py", "X25519", 1001), ("monty.py", "NIST256", 1002), ("monty.py", "X448", 1003)):
        fx, ref = field_fixture(script, prime, seed)
        path = gio.dump(fx, "field_%s.json" % prime)
        print(prime, "time:", {k: v for k, v in fx["time"].items() if "check" in k})
        out[prime] = path
    for script, prime, seed in (("pseudo.py", "X25519", 2001), ("monty.py", "X448", 2003)):
        fx = lazy_fixture(script, prime, seed)
        gio.dump(fx, "field_%s_lazy.json" % prime)
    for script, prime, seed in (("pseudo.py", "X25519", 4001), ("monty.py", "NIST256", 4002), ("monty.py", "X448", 4003)):
        gio.dump(sqrt_fixture(script, prime, seed), "sqrt_%s.json" % prime)
    extras()
    for k, wname in enumerate(("NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W")):
        gio.dump(weierstrass_fixture(wname, 8001 + k, pairs=32 if k == 0 else (24, 20, 28, 28)[k - 1]), "weierstrass_%s.json" % wname)
    for name, seed in (("ED25519", 5001), ("ED448", 5002), ("NUMS256E", 5003), ("ED248", 5004), ("ED376", 5005), ("ED500", 5006)):
        gio.dump(edwards_fixture(name, seed, pairs={"ED25519": 40, "ED376": 16, "ED500": 12}.get(name, 20)), "edwards_%s.json" % name)
    for curve, seed in (("X25519", 3001), ("X448", 3003)):
        fx = ladder_fixture(curve, seed)
        gio.dump(fx, "ladder_%s.json" % curve)
        print(curve, "ref main chain 5000:", fx["ref_main_chain"]["checkpoints"]["5000"], "dh:", fx["ref_main_chain"]["dh"]["shared"])


if __name__ == "__main__":
    main()
