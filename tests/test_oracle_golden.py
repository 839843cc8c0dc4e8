"""The CPU oracle against the golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU-only; this is what pins the oracle."""
import ctypes

import pytest

from tests.conftest import limbs, load_golden
from tests.oracle_binding import PRIMES

ALL = ["X25519", "NIST256", "X448"]


@pytest.fixture(scope="module", params=ALL)
def fx(request):
    return request.param, load_golden("field_%s.json" % request.param)


def test_macro_block(fx):
    P, g = fx
    N, radix, nbits, nbytes = PRIMES[P]
    hdr = " ".join(g["params"]["header"])
    assert "#define Nlimbs %d" % N in hdr and "#define Radix %d" % radix in hdr
    assert "#define Nbits %d" % nbits in hdr and "#define Nbytes %d" % nbytes in hdr


@pytest.mark.parametrize("op", ["modadd", "modsub", "modmul"])
def test_binary_ops(oracle, fx, op):
    P, g = fx
    for a, b, want in zip(g["A"], g["B"], g["ops"][op]):
        assert oracle.bi(op, P, limbs(a), limbs(b)) == limbs(want)


@pytest.mark.parametrize("op", ["modneg", "modsqr", "redc", "nres"])
def test_unary_ops(oracle, fx, op):
    P, g = fx
    for a, want in zip(g["A"], g["ops"][op]):
        assert oracle.un(op, P, limbs(a)) == limbs(want)


def test_nres_of_raw(oracle, fx):
    P, g = fx
    for a, want in zip(g["raw_a"], g["nres_raw_a"]):
        assert oracle.un("nres", P, limbs(a)) == limbs(want)


def test_chained_noncanonical(oracle, fx):
    P, g = fx
    o = g["ops"]
    for i, (a, b) in enumerate(zip(g["A"], g["B"])):
        a, b = limbs(a), limbs(b)
        C = oracle.bi("modmul", P, a, b)
        D = oracle.bi("modsub", P, a, b)
        E = oracle.bi("modadd", P, a, b)
        assert oracle.bi("modmul", P, C, D) == limbs(o["chain_mul_CD"][i])
        assert oracle.un("modsqr", P, D) == limbs(o["chain_sqr_D"][i])
        assert oracle.bi("modadd", P, D, E) == limbs(o["chain_add_DE"][i])
        assert oracle.bi("modsub", P, E, C) == limbs(o["chain_sub_EC"][i])
        assert oracle.un("redc", P, C) == limbs(o["chain_redc_C"][i])


def test_aliasing(oracle, fx):
    P, g = fx
    for a, b, want in zip(g["A"], g["B"], g["ops"]["alias_chain"]):
        x, y = oracle.arr(P, limbs(a)), oracle.arr(P, limbs(b))
        oracle.fn("modmul", P)(x, y, x)
        oracle.fn("modsqr", P)(x, x)
        oracle.fn("modadd", P)(x, x, x)
        oracle.fn("modsub", P)(x, y, x)
        assert list(x) == limbs(want)


def test_modmli(oracle, fx):
    P, g = fx
    ints = g["ops"]["modmli_ints"]
    for a, row in zip(g["A"], g["ops"]["modmli"]):
        for k, want in zip(ints, row):
            z = oracle.arr(P)
            oracle.fn("modmli", P)(oracle.arr(P, limbs(a)), k, z)
            assert list(z) == limbs(want)


def test_modfsb_flatten(oracle, fx):
    P, g = fx
    for f in ("modfsb", "flatten"):
        for a, (want, ret) in zip(g["A"], g["ops"][f]):
            x = oracle.arr(P, limbs(a))
            r = oracle.fn(f, P)(x)
            assert list(x) == limbs(want) and int(r) == ret


def test_predicates(oracle, fx):
    P, g = fx
    o = g["ops"]
    for i, (a, b) in enumerate(zip(g["A"], g["B"])):
        x, y = oracle.arr(P, limbs(a)), oracle.arr(P, limbs(b))
        assert oracle.fn("modis1", P)(x) == o["modis1"][i]
        assert oracle.fn("modis0", P)(x) == o["modis0"][i]
        assert oracle.fn("modsign", P)(x) == o["modsign"][i]
        assert oracle.fn("modcmp", P)(x, y) == o["modcmp"][i]
    assert any(o["modis0"]) and any(o["modis1"])  # the corner pool really hits both


def test_modhaf_shifts(oracle, fx):
    P, g = fx
    for a, want in zip(g["A"], g["ops"]["modhaf"]):
        x = oracle.arr(P, limbs(a))
        oracle.fn("modhaf", P)(x)
        assert list(x) == limbs(want)
    for a, rec in zip(g["A"], g["ops"]["shifts"]):
        x = oracle.arr(P, oracle.un("redc", P, limbs(a)))
        oracle.fn("modshl", P)(rec["k"], x)
        assert list(x) == limbs(rec["shl_of_redc"])
        x = oracle.arr(P, limbs(a))
        r = oracle.fn("modshr", P)(rec["k"], x)
        assert list(x) == limbs(rec["shr"]) and r == rec["shr_ret"]


def test_cond_move_swap(oracle, fx):
    P, g = fx
    for a, b, rec in zip(g["A"], g["B"], g["ops"]["cond"]):
        x, y = oracle.arr(P, limbs(a)), oracle.arr(P, limbs(b))
        oracle.fn("modcsw", P)(rec["d"], x, y)
        assert list(x) == limbs(rec["csw_g"]) and list(y) == limbs(rec["csw_f"])
        y = oracle.arr(P, limbs(b))
        oracle.fn("modcmv", P)(rec["d"], oracle.arr(P, limbs(a)), y)
        assert list(y) == limbs(rec["cmv_f"])


def test_constants(oracle, fx):
    P, g = fx
    o = g["ops"]
    x = oracle.arr(P); oracle.fn("modone", P)(x); assert list(x) == limbs(o["modone"])
    x = oracle.arr(P, [7] * PRIMES[P][0]); oracle.fn("modzer", P)(x); assert list(x) == limbs(o["modzer"])
    for k, want in o["modint"]:
        x = oracle.arr(P); oracle.fn("modint", P)(k, x); assert list(x) == limbs(want)
    for k, want in o["mod2r"]:
        x = oracle.arr(P); oracle.fn("mod2r", P)(k, x); assert list(x) == limbs(want)


def test_bytes_io(oracle, fx):
    P, g = fx
    nb = PRIMES[P][3]
    for rec in g["ops"]["bytes"]:
        x = oracle.arr(P)
        r = oracle.fn("modimp", P)(bytes.fromhex(rec["bytes"]), x)
        assert list(x) == limbs(rec["imp"]) and r == rec["imp_ret"]
        out = ctypes.create_string_buffer(nb)
        oracle.fn("modexp", P)(x, out)
        assert out.raw.hex() == rec["exp"]
    for a, want in zip(g["A"], g["ops"]["modexp_A"]):
        out = ctypes.create_string_buffer(nb)
        oracle.fn("modexp", P)(oracle.arr(P, limbs(a)), out)
        assert out.raw.hex() == want


def test_modinv_after_redc(oracle, fx):
    """modpro/modinv use our own addition chain, so only the canonical value is comparable
    (SURVEY 8(c) caveat 1); modinv(0) = 0."""
    P, g = fx
    for rec in g["ops"]["modinv"]:
        z = oracle.arr(P)
        oracle.fn("modinv", P)(oracle.arr(P, limbs(rec["x"])), None, z)
        assert oracle.un("redc", P, list(z)) == limbs(rec["inv_redc"])
        # progenitor path: h = modpro(x) supplied by the caller (rfc7748.c:226-227)
        h = oracle.arr(P)
        oracle.fn("modpro", P)(oracle.arr(P, limbs(rec["x"])), h)
        z2 = oracle.arr(P)
        oracle.fn("modinv", P)(oracle.arr(P, limbs(rec["x"])), h, z2)
        assert list(z2) == list(z)


def test_modsqrt_modqr(oracle, fx):
    """modsqrt/modqr against the value-level pins (tests/golden/sqrt_*.json) and, for every input,
    the progenitor-supplied form (edwards.c-style callers pass h = modpro(x))."""
    P, _ = fx
    g = load_golden("sqrt_%s.json" % P)
    for rec in g["recs"]:
        x = limbs(rec["x"])
        r = oracle.arr(P)
        oracle.fn("modsqrt", P)(oracle.arr(P, x), None, r)
        assert oracle.un("redc", P, list(r)) == limbs(rec["sqrt_redc"])
        assert oracle.fn("modqr", P)(None, oracle.arr(P, x)) == rec["qr"]
        h = oracle.arr(P)
        oracle.fn("modpro", P)(oracle.arr(P, x), h)
        r2 = oracle.arr(P)
        oracle.fn("modsqrt", P)(oracle.arr(P, x), h, r2)
        assert list(r2) == list(r) and oracle.fn("modqr", P)(h, oracle.arr(P, x)) == rec["qr"]


def test_reference_selftest_chain(oracle, fx):
    """the generators' own acceptance test (pseudo.py:1762-1855): for random x,y < 2p the chain
    nres,nres,modadd,modsub,modmul,modsqr,modinv,modsqrt,modsqr,modhaf,modadd,modshl,modshr,redc
    must give inverse(((x-y)(x+y))^2 mod p)."""
    import random
    from modarith_amd.params import derive
    P, _ = fx
    fp = derive(P)
    rng = random.Random(2024)
    f = lambda name: oracle.fn(name, P)
    for _ in range(200):
        x, y = rng.randrange(0, 2 * fp.p), rng.randrange(0, 2 * fp.p)
        want = pow(((x - y) * (x + y)) ** 2 % fp.p, -1, fp.p) if ((x - y) * (x + y)) % fp.p else 0
        ax, ay, at, az = oracle.arr(P, fp.to_limbs(x)), oracle.arr(P, fp.to_limbs(y)), oracle.arr(P), oracle.arr(P)
        f("nres")(ax, ax); f("nres")(ay, ay)
        f("modadd")(ax, ay, at); f("modsub")(ax, ay, az)
        f("modmul")(at, az, ax); f("modsqr")(ax, az)
        f("modinv")(az, None, az)
        f("modsqrt")(az, None, az); f("modsqr")(az, az)
        f("modhaf")(az); f("modadd")(az, az, az)
        f("modshl")(1, az); f("modshr")(1, az)
        f("redc")(az, az)
        assert fp.from_limbs(list(az)) == want


def test_time_protocol_check_words(oracle, fx):
    """time.c protocol (pseudo.py:1235-1250): 1k- and 100k-deep prefixes of the reference chains here;
    the full 10^8-deep run is `oracle/time_oracle` (bench.py runs it as the cpu_baseline)."""
    P, g = fx
    t = g["time"]
    N, radix = PRIMES[P][0], PRIMES[P][1]
    mk = lambda v: [(int(v, 16) >> (radix * i)) & ((1 << radix) - 1) for i in range(N)]
    for outer, tag in ((1, "1k"), (100, "100k")):
        x, y = oracle.arr(P, mk(t["ra"])), oracle.arr(P, mk(t["rb"]))
        assert oracle.fn("time_modmul", P)(x, y, outer) == int(t["modmul_check_" + tag], 16)
        assert list(x) == limbs(t["modmul_z_" + tag])
        x = oracle.arr(P, mk(t["rs"]))
        assert oracle.fn("time_modsqr", P)(x, outer) == int(t["modsqr_check_" + tag], 16)
        assert list(x) == limbs(t["modsqr_z_" + tag])
    x = oracle.arr(P, mk(t["ri"]))
    assert oracle.fn("time_modinv", P)(x, 3) == int(t["modinv_check_full"], 16)
    assert list(x) == limbs(t["modinv_z_full"])


@pytest.mark.parametrize("P", ["X25519", "X448"])
def test_lazy_forms(oracle, P):
    g = load_golden("field_%s_lazy.json" % P)
    for a, b, wa, ws, wn in zip(g["A"], g["B"], g["ops"]["modadd"], g["ops"]["modsub"], g["ops"]["modneg"]):
        a, b = limbs(a), limbs(b)
        assert oracle.bi("modadd_lazy", P, a, b) == limbs(wa)
        assert oracle.bi("modsub_lazy", P, a, b) == limbs(ws)
        assert oracle.un("modneg_lazy", P, a) == limbs(wn)


@pytest.mark.parametrize("C", ["X25519", "X448"])
def test_ladder_kats(oracle, C):
    g = load_golden("ladder_%s.json" % C)
    for rec in g["kat"] + g["pairs"]:
        assert oracle.ladder(C, bytes.fromhex(rec["k"]), bytes.fromhex(rec["u"])).hex() == rec["out"], rec
    # aliasing bv == bu (rfc7748.c:329)
    nb = PRIMES[C][3]
    rec = g["kat"][0]
    buf = ctypes.create_string_buffer(bytes.fromhex(rec["u"]), nb)
    getattr(oracle.lib, "rfc7748_" + C)(bytes.fromhex(rec["k"]), buf, buf)
    assert buf.raw.hex() == rec["out"]


@pytest.mark.parametrize("C,iters", [("X25519", 100), ("X448", 10)])
def test_ladder_reference_main_chain(oracle, C, iters):
    """the reference main()'s LCG-keyed chain (rfc7748.c:297-305), prefix checkpoints + DH block."""
    g = load_golden("ladder_%s.json" % C)["ref_main_chain"]
    bk, bu = bytes.fromhex(g["bk"]), bytes.fromhex(g["bu0"])
    for i in range(iters):
        bv = oracle.ladder(C, bk, bu)
        bu = oracle.ladder(C, bk, bv)
        if str(i + 1) in g["checkpoints"]:
            assert bu.hex() == g["checkpoints"][str(i + 1)]
    dh = g["dh"]
    assert oracle.ladder(C, bytes.fromhex(dh["alice"]), bytes.fromhex(dh["bpk"])).hex() == dh["shared"]
    assert oracle.ladder(C, bytes.fromhex(dh["bob"]), bytes.fromhex(dh["apk"])).hex() == dh["shared"]


def test_ladder_iteration_rfc(oracle):
    g = load_golden("ladder_X25519.json")["iter"]
    k = u = bytes.fromhex(g["start"])
    for i in range(1000):
        k, u = oracle.ladder("X25519", k, u), k
        if i == 0:
            assert k.hex() == g["after_1"]
    assert k.hex() == g["after_1000"]


# ---- round-2 fixtures (tests/golden/make_golden_r2.py): modnsqr and OUT-OF-CONTRACT limbs from the reference
def test_modnsqr_golden(oracle, fx):
    P, _ = fx
    for rec in load_golden("field_%s_r2.json" % P)["modnsqr"]:
        z = oracle.arr(P, limbs(rec["a"]))
        oracle.fn("modnsqr", P)(z, rec["k"])
        assert list(z) == limbs(rec["out"]), (P, rec["k"])


def test_out_of_contract_limbs_golden(oracle, fx):
    """field.c has no error path: limbs beyond the excess budget wrap in 64 bits.  The oracle restates exactly
    that arithmetic, so it must reproduce the reference's answers on such inputs too (this is what lets it judge
    the engine's exact-product fallback and the mixed-wave policy switch)."""
    P, _ = fx
    for i, rec in enumerate(load_golden("field_%s_r2.json" % P)["ooc"]):
        a, b = limbs(rec["a"]), limbs(rec["b"])
        for op in ("modmul", "modadd", "modsub"):
            assert oracle.bi(op, P, a, b) == limbs(rec[op]), (P, op, i)
        for op in ("modsqr", "nres", "redc", "modneg"):
            assert oracle.un(op, P, a) == limbs(rec[op]), (P, op, i)
        z = oracle.arr(P)
        oracle.fn("modmli", P)(oracle.arr(P, a), 121665, z)
        assert list(z) == limbs(rec["modmli_121665"]), (P, "modmli", i)


# ---- round-3 fixture (tests/golden/make_bulk_digests.py): 2^18 elements per input class through the REFERENCE's emitted
# modmul / modsqr / modadd / modsub / modneg / nres / redc / modmli, committed as one sha256 digest per 4096-element block
@pytest.mark.parametrize("P", ["X25519", "NIST256", "X448"])
def test_bulk_digests_oracle(oracle, P):
    """the oracle reproduces the reference on 3 x 2^18 elements per prime and operation (SURVEY.md:359)"""
    from tests.util import BULK_CLASSES, BULK_OPS, block_digests, bulk_inputs, oracle_bin, oracle_mli, oracle_un
    g = load_golden("bulk_digests.json")
    for cls in BULK_CLASSES:
        a, b = bulk_inputs(P, cls, g["n"])
        for op in BULK_OPS:
            if op in ("modmul", "modadd", "modsub"):
                c = oracle_bin(oracle, op, P, a, b)
            elif op == "modmli_121665":
                c = oracle_mli(oracle, P, a, 121665)
            else:
                c = oracle_un(oracle, op, P, a)
            assert block_digests(c, g["block"]) == g["primes"][P][cls][op], (P, cls, op)


# ---- the reference's corner-case protocol (edge.py / edge.c), restated in tests/edge_protocol.py
class _OracleEdgeEngine:
    def __init__(self, oracle, P, nbytes):
        self.o, self.P, self.nb = oracle, P, nbytes
    def imp(self, ints):
        out = []
        for v in ints:
            x = self.o.arr(self.P)
            self.o.fn("modimp", self.P)(int(v).to_bytes(self.nb, "big"), x)
            out.append(list(x))
        return out
    def _un(self, f, xs):
        return [self.o.un(f, self.P, x) for x in xs]
    def _bi(self, f, xs, ys):
        return [self.o.bi(f, self.P, x, y) for x, y in zip(xs, ys)]
    def inv(self, xs):
        out = []
        for x in xs:
            z = self.o.arr(self.P)
            self.o.fn("modinv", self.P)(self.o.arr(self.P, x), None, z)
            out.append(list(z))
        return out
    def sqrt(self, xs):
        out = []
        for x in xs:
            z = self.o.arr(self.P)
            self.o.fn("modsqrt", self.P)(self.o.arr(self.P, x), None, z)
            out.append(list(z))
        return out
    def add(self, xs, ys): return self._bi("modadd", xs, ys)
    def sub(self, xs, ys): return self._bi("modsub", xs, ys)
    def mul(self, xs, ys): return self._bi("modmul", xs, ys)
    def sqr(self, xs): return self._un("modsqr", xs)
    def cmp(self, xs, ys):
        return [self.o.fn("modcmp", self.P)(self.o.arr(self.P, x), self.o.arr(self.P, y)) for x, y in zip(xs, ys)]


@pytest.mark.parametrize("P", ["X25519", "NIST256", "X448", "NIST384", "NIST521", "SECP256K1", "NUMS256W", "ED248", "ED376", "ED500"])
def test_edge_protocol_oracle(oracle, P):
    """edge.py's 17 corner pairs and their doubled forms through the oracle: 1/(1/a), a+b, a-b, b-a, a*b, sqr(sqrt(sqr a))"""
    from modarith_amd.params import derive
    from tests import edge_protocol
    fp = derive(P)
    assert edge_protocol.run(_OracleEdgeEngine(oracle, P, fp.nbytes), fp.p, fp.n, fp.nbytes) == []
