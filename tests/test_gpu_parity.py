"""GPU parity: the HIP kernels, called through the C-ABI, against (1) the golden vectors produced by
the reference and (2) the CPU oracle on seeded batches.  Bit-exact limbs everywhere (integer work);
modpro/modinv are compared after redc (different addition chain, SURVEY 8(c) caveat 1)."""
import ctypes

import numpy as np
import pytest

from tests.conftest import limbs, load_golden
from tests.oracle_binding import PRIMES
from tests.util import oracle_bin, oracle_mli, oracle_un, random_soa, to_dev, to_np, vp

pytestmark = pytest.mark.gpu
ALL = ["X25519", "NIST256", "X448"]                      # BASELINE.json configs: golden vectors AND oracle batches
EXTRA = list(__import__("modarith_amd.emit", fromlist=["EXTRA_PRIMES"]).EXTRA_PRIMES) + __import__("tests.util", fromlist=["generated_tags"]).generated_tags()   # further primes: golden vectors + generic oracle


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module", params=ALL + EXTRA)
def ctx(request, torch_cuda):
    from modarith_amd.field import Field
    P = request.param
    return P, Field(P), load_golden("field_%s.json" % P)


def dev(F, rows):
    return F.from_limbs([limbs(r) for r in rows])


def rows_of(F, t):
    return F.to_limbs(t)


# ---------------------------------------------------------------- against the reference's own outputs
@pytest.mark.parametrize("op", ["modadd", "modsub", "modmul"])
def test_golden_binary(ctx, op):
    P, F, g = ctx
    got = rows_of(F, getattr(F, op)(dev(F, g["A"]), dev(F, g["B"])))
    assert got == [limbs(r) for r in g["ops"][op]]


@pytest.mark.parametrize("op", ["modneg", "modsqr", "redc", "nres"])
def test_golden_unary(ctx, op):
    P, F, g = ctx
    got = rows_of(F, getattr(F, op)(dev(F, g["A"])))
    assert got == [limbs(r) for r in g["ops"][op]]


def test_golden_chained_and_alias(ctx):
    P, F, g = ctx
    A, B = dev(F, g["A"]), dev(F, g["B"])
    C, D, E = F.modmul(A, B), F.modsub(A, B), F.modadd(A, B)
    o = g["ops"]
    assert rows_of(F, F.modmul(C, D)) == [limbs(r) for r in o["chain_mul_CD"]]
    assert rows_of(F, F.modsqr(D)) == [limbs(r) for r in o["chain_sqr_D"]]
    assert rows_of(F, F.modadd(D, E)) == [limbs(r) for r in o["chain_add_DE"]]
    assert rows_of(F, F.modsub(E, C)) == [limbs(r) for r in o["chain_sub_EC"]]
    assert rows_of(F, F.redc(C)) == [limbs(r) for r in o["chain_redc_C"]]
    # in-place aliasing, as the reference uses it (pseudo.py:752,1783,1793; rfc7748.c:214)
    k = len(o["alias_chain"])
    x, y = dev(F, g["A"][:k]), dev(F, g["B"][:k])
    F.modmul(x, y, out=x)
    F.modsqr(x, out=x)
    F.modadd(x, x, out=x)
    F.modsub(x, y, out=x)
    assert rows_of(F, x) == [limbs(r) for r in o["alias_chain"]]


def test_golden_modmli(ctx):
    P, F, g = ctx
    ints = g["ops"]["modmli_ints"]
    rows = g["ops"]["modmli"]
    A = dev(F, g["A"][:len(rows)])
    for i, k in enumerate(ints):
        assert rows_of(F, F.modmli(A, k)) == [limbs(r[i]) for r in rows]


def test_golden_modfsb_flatten_predicates(ctx):
    P, F, g = ctx
    o = g["ops"]
    for fn in ("modfsb", "flatten"):
        x = dev(F, g["A"])
        flag = getattr(F, fn)(x)
        assert rows_of(F, x) == [limbs(r[0]) for r in o[fn]]
        assert flag.cpu().tolist() == [r[1] for r in o[fn]]
    A, B = dev(F, g["A"]), dev(F, g["B"])
    assert F.modis1(A).cpu().tolist() == o["modis1"]
    assert F.modis0(A).cpu().tolist() == o["modis0"]
    assert F.modsign(A).cpu().tolist() == o["modsign"]
    assert F.modcmp(A, B).cpu().tolist() == o["modcmp"]
    x = dev(F, g["A"])
    F.modhaf(x)
    assert rows_of(F, x) == [limbs(r) for r in o["modhaf"]]


def test_golden_shifts_cond_consts(ctx, torch_cuda):
    torch = torch_cuda
    P, F, g = ctx
    o = g["ops"]
    for i, rec in enumerate(o["shifts"][:24]):
        a = dev(F, [g["A"][i]])
        x = F.redc(a)
        F.modshl(rec["k"], x)
        assert rows_of(F, x) == [limbs(rec["shl_of_redc"])]
        y = dev(F, [g["A"][i]])
        r = F.modshr(rec["k"], y)
        assert rows_of(F, y) == [limbs(rec["shr"])] and r.cpu().tolist() == [rec["shr_ret"]]
    k = len(o["cond"])
    d = torch.tensor([r["d"] for r in o["cond"]], dtype=torch.int32, device="cuda")
    gq, fq = dev(F, g["A"][:k]), dev(F, g["B"][:k])
    F.modcsw(d, gq, fq)
    assert rows_of(F, gq) == [limbs(r["csw_g"]) for r in o["cond"]]
    assert rows_of(F, fq) == [limbs(r["csw_f"]) for r in o["cond"]]
    f2 = dev(F, g["B"][:k])
    F.modcmv(d, dev(F, g["A"][:k]), f2)
    assert rows_of(F, f2) == [limbs(r["cmv_f"]) for r in o["cond"]]
    assert rows_of(F, F.modone(3)) == [limbs(o["modone"])] * 3
    assert rows_of(F, F.modzer(2)) == [limbs(o["modzer"])] * 2
    for kk, want in o["modint"]:
        assert rows_of(F, F.modint(kk, 1)) == [limbs(want)]
    for kk, want in o["mod2r"]:
        assert rows_of(F, F.mod2r(kk, 1)) == [limbs(want)]


def test_golden_bytes(ctx, torch_cuda):
    torch = torch_cuda
    P, F, g = ctx
    recs = g["ops"]["bytes"]
    b = torch.tensor([list(bytes.fromhex(r["bytes"])) for r in recs], dtype=torch.uint8, device="cuda")
    a, flag = F.modimp(b)
    assert rows_of(F, a) == [limbs(r["imp"]) for r in recs]
    assert flag.cpu().tolist() == [r["imp_ret"] for r in recs]
    out = F.modexp(a).cpu().numpy()
    assert [bytes(row).hex() for row in out] == [r["exp"] for r in recs]
    k = len(g["ops"]["modexp_A"])
    out = F.modexp(dev(F, g["A"][:k])).cpu().numpy()
    assert [bytes(row).hex() for row in out] == g["ops"]["modexp_A"]


def test_golden_modinv_after_redc(ctx):
    P, F, g = ctx
    recs = g["ops"]["modinv"]
    x = dev(F, [r["x"] for r in recs])
    z = F.modinv(x)
    assert rows_of(F, F.redc(z)) == [limbs(r["inv_redc"]) for r in recs]
    h = F.modpro(x)
    z2 = F.modinv(x, h)
    assert rows_of(F, z2) == rows_of(F, z)


def test_golden_modsqrt_modqr(ctx):
    P, F, _ = ctx
    recs = load_golden("sqrt_%s.json" % P)["recs"]
    x = dev(F, [r["x"] for r in recs])
    root = F.modsqrt(x)
    assert rows_of(F, F.redc(root)) == [limbs(r["sqrt_redc"]) for r in recs]
    assert F.modqr(None, x).cpu().tolist() == [r["qr"] for r in recs]
    h = F.modpro(x)
    assert rows_of(F, F.modsqrt(x, h)) == rows_of(F, root)
    assert F.modqr(h, x).cpu().tolist() == [r["qr"] for r in recs]


# ---------------------------------------------------------------- against the oracle, seeded batches
N_BIG = (1 << 16) + 3  # odd on purpose: exercises the 16-byte path plus its scalar tail


@pytest.mark.parametrize("P", ALL)
def test_oracle_batch_all_ops(oracle, torch_cuda, P):
    from modarith_amd.field import Field
    F = Field(P)
    a, b = random_soa(P, N_BIG, 11), random_soa(P, N_BIG, 12)
    da, db = to_dev(a), to_dev(b)
    for op in ("modmul", "modadd", "modsub", "modadd_lazy", "modsub_lazy"):
        assert np.array_equal(to_np(getattr(F, op)(da, db)), oracle_bin(oracle, op, P, a, b)), op
    for op in ("modsqr", "modneg", "modneg_lazy", "nres", "redc", "modcpy"):
        assert np.array_equal(to_np(getattr(F, op)(da)), oracle_un(oracle, op, P, a)), op
    for k in (121665, 39081, 1, 0x7fffffff, -3):
        assert np.array_equal(to_np(F.modmli(da, k)), oracle_mli(oracle, P, a, k)), k
    # shared multiplicand
    b0 = [int(v) for v in b[:, 5]]
    want = np.empty_like(a)
    b0a = (ctypes.c_uint64 * len(b0))(*b0)
    oracle.fn("batch_modmul_shared", P)(vp(a), b0a, vp(want), a.shape[1], a.shape[1])
    assert np.array_equal(to_np(F.modmuls(da, b0)), want)


@pytest.mark.parametrize("P", ALL)
def test_oracle_unaligned_and_strided(oracle, torch_cuda, P):
    """odd offsets / sub-batch views: the scalar-width path and a limb stride larger than n."""
    import torch
    from modarith_amd.field import Field
    F = Field(P)
    n = 1000
    a, b = random_soa(P, n + 7, 21), random_soa(P, n + 7, 22)
    da, db = to_dev(a), to_dev(b)
    for off, cnt in ((1, 999), (0, 1), (3, 2), (2, 501), (0, n + 7)):
        va, vb = da[:, off:off + cnt], db[:, off:off + cnt]
        out = torch.zeros_like(da)
        F.modmul(va, vb, out=out[:, off:off + cnt])
        want = oracle_bin(oracle, "modmul", P, np.ascontiguousarray(a[:, off:off + cnt]), np.ascontiguousarray(b[:, off:off + cnt]))
        got = to_np(out)
        assert np.array_equal(got[:, off:off + cnt], want)
        assert not got[:, :off].any() and not got[:, off + cnt:].any()  # nothing outside the slice is touched
    assert F.modmul(da[:, :0], db[:, :0]).shape[1] == 0  # empty batch


@pytest.mark.parametrize("P", ALL)
def test_oracle_modinv_batch(oracle, torch_cuda, P):
    from modarith_amd.field import Field
    F = Field(P)
    a = random_soa(P, 4099, 31)
    z = F.redc(F.modinv(to_dev(a)))
    want = oracle_un(oracle, "redc", P, oracle_un(oracle, "modinv", P, a))
    assert np.array_equal(to_np(z), want)
    r = F.redc(F.modsqrt(to_dev(a)))
    want = oracle_un(oracle, "redc", P, oracle_un(oracle, "modsqrt", P, a))
    assert np.array_equal(to_np(r), want)
    q = np.empty(a.shape[1], dtype=np.int32)
    oracle.fn("batch_modqr", P)(vp(a), vp(q), a.shape[1], a.shape[1])
    assert np.array_equal(F.modqr(None, to_dev(a)).cpu().numpy(), q)


def test_properties_full_size(torch_cuda):
    """BASELINE config 2 size (2^24 elements, 5x51): size-independent properties on the GPU alone:
    (a*b)*c == a*(b*c), a*(b+c) == a*b + a*c, a*a == sqr(a), a * 1/a == 1 (all after redc)."""
    torch = torch_cuda
    from modarith_amd.field import Field
    F = Field("X25519")
    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(7)
    def rnd():
        t = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda", generator=g)
        return t
    a, b, c = rnd(), rnd(), rnd()
    lhs = F.redc(F.modmul(F.modmul(a, b), c))
    rhs = F.redc(F.modmul(a, F.modmul(b, c)))
    assert torch.equal(lhs, rhs)
    lhs = F.redc(F.modmul(a, F.modadd(b, c)))
    rhs = F.redc(F.modadd(F.modmul(a, b), F.modmul(a, c)))
    assert torch.equal(lhs, rhs)
    assert torch.equal(F.modsqr(a), F.modmul(a, a))
    m = 1 << 18
    x = a[:, :m].contiguous()
    one = F.redc(F.modmul(x, F.modinv(x)))
    assert int(F.modis1(one).sum()) + int(F.modis0(x).sum()) == m


# ---------------------------------------------------------------- scalar (_ct) form: the reference's own self-test shape
@pytest.mark.parametrize("P", ALL)
def test_scalar_abi_reference_selftest_chain(oracle, torch_cuda, P):
    """The generators' ctypes self-test (pseudo.py:1783-1796), the whole chain: nres, nres, modadd,
    modsub, modmul, modsqr, modinv, modsqrt, modsqr, modhaf, modadd, modshl, modshr, redc
    == inverse(((x-y)(x+y))^2), through the scalar entry points with the reference's signatures."""
    import random
    from modarith_amd import _lib
    from tests.util import derive_any as derive
    lib = _lib.load()
    fp = derive(P)
    N = fp.nlimbs
    U = ctypes.c_uint64 * N
    f = lambda name: getattr(lib, "%s_%s_ct" % (name, P))
    rng = random.Random(99)
    for _ in range(8):
        x, y = rng.randrange(0, 2 * fp.p), rng.randrange(0, 2 * fp.p)
        want = pow(((x - y) * (x + y)) ** 2 % fp.p, -1, fp.p)
        ax, ay, at, az = U(*fp.to_limbs(x)), U(*fp.to_limbs(y)), U(), U()
        f("nres")(ax, ax); f("nres")(ay, ay)
        f("modadd")(ax, ay, at); f("modsub")(ax, ay, az)
        f("modmul")(at, az, ax); f("modsqr")(ax, az)
        f("modinv")(az, None, az)
        f("modsqrt")(az, None, az); f("modsqr")(az, az)
        f("modhaf")(az); f("modadd")(az, az, az)
        f("modshl")(1, az); f("modshr")(1, az)
        f("redc")(az, az)
        assert fp.from_limbs(list(az)) == want


# ---------------------------------------------------------------- time.c protocol on the device (row a15)
@pytest.mark.parametrize("P", ALL)
def test_time_protocol_check_words(torch_cuda, P):
    """every lane runs the reference's dependent chains (pseudo.py:1235-1250) on the seed-42 operands;
    the 24-bit check words and redc'd limbs must equal the reference's (golden "time", depth 10^3 / 10^5)."""
    from modarith_amd.field import Field
    F = Field(P)
    g = load_golden("field_%s.json" % P)["time"]
    N, radix = PRIMES[P][0], PRIMES[P][1]
    mk = lambda v: [(int(v, 16) >> (radix * i)) & ((1 << radix) - 1) for i in range(N)]
    lanes = 130
    x, y = F.from_limbs([mk(g["ra"])] * lanes), F.from_limbs([mk(g["rb"])] * lanes)
    xs, xi = F.from_limbs([mk(g["rs"])] * lanes), F.from_limbs([mk(g["ri"])] * lanes)
    for outer, tag in ((1, "1k"), (100, "100k")):
        z = F.to_limbs(F.time_protocol("modmul", x, y, outer))
        assert z == [limbs(g["modmul_z_" + tag])] * lanes and z[0][0] & 0xFFFFFF == int(g["modmul_check_" + tag], 16)
        z = F.to_limbs(F.time_protocol("modsqr", xs, None, outer))
        assert z == [limbs(g["modsqr_z_" + tag])] * lanes and z[0][0] & 0xFFFFFF == int(g["modsqr_check_" + tag], 16)
    z = F.to_limbs(F.time_protocol("modinv", xi, None, 2))
    assert z == [limbs(g["modinv_z_full"])] * lanes and z[0][0] & 0xFFFFFF == int(g["modinv_check_full"], 16)


# ---------------------------------------------------------------- ladder
@pytest.mark.parametrize("C", ["X25519", "X448"])
def test_ladder_kats_and_pairs(torch_cuda, C):
    torch = torch_cuda
    from modarith_amd.field import rfc7748
    g = load_golden("ladder_%s.json" % C)
    recs = g["kat"] + g["pairs"]
    k = torch.tensor([list(bytes.fromhex(r["k"])) for r in recs], dtype=torch.uint8, device="cuda")
    u = torch.tensor([list(bytes.fromhex(r["u"])) for r in recs], dtype=torch.uint8, device="cuda")
    out = rfc7748(C, k, u).cpu().numpy()
    assert [bytes(row).hex() for row in out] == [r["out"] for r in recs]
    # aliasing bv == bu (rfc7748.c:329)
    u2 = u.clone()
    rfc7748(C, k, u2, out=u2)
    assert [bytes(row).hex() for row in u2.cpu().numpy()] == [r["out"] for r in recs]


@pytest.mark.parametrize("C,n", [("X25519", 8192 + 5), ("X448", 2048 + 3)])
def test_ladder_vs_oracle_random(oracle, torch_cuda, C, n):
    torch = torch_cuda
    from modarith_amd.field import rfc7748
    nb = PRIMES[C][3]
    rng = np.random.default_rng(5)
    k = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    u = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    u[0] = 0; u[1] = 255; u[2, :] = 0; u[2, 0] = 9
    want = np.empty_like(u)
    oracle.lib.oracle_parallel(3 if C == "X25519" else 4, vp(k), vp(u), vp(want), n, 0, 8)
    got = rfc7748(C, torch.from_numpy(k).cuda(), torch.from_numpy(u).cuda()).cpu().numpy()
    assert np.array_equal(got, want)


def test_ladder_scalar_abi_chain(torch_cuda):
    """reference main()'s chain shape (rfc7748.c:297-305) through the scalar entry point, 10 rounds."""
    from modarith_amd import _lib
    lib = _lib.load()
    g = load_golden("ladder_X25519.json")["ref_main_chain"]
    bk = bytes.fromhex(g["bk"])
    bu = ctypes.create_string_buffer(bytes.fromhex(g["bu0"]), 32)
    bv = ctypes.create_string_buffer(32)
    for i in range(10):
        lib.rfc7748_X25519(bk, bu, bv)
        lib.rfc7748_X25519(bk, bv, bu)
    assert bu.raw.hex() == g["checkpoints"]["10"]


def test_c_drop_in_example(torch_cuda, tmp_path):
    """examples/rfc7748_drop_in.c: the reference main()'s sequence (rfc7748.c:259-341) in plain C against
    the library; stdout must carry the RFC vector, the reference chain checkpoint and DH secret."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rfc7748_drop_in")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "rfc7748_drop_in.c"), "-I", os.path.join(root, "include"),
                           "-L", os.path.join(root, "modarith_amd"), "-l:libmodarith_amd.so",
                           "-Wl,-rpath," + os.path.join(root, "modarith_amd"), "-o", exe])
    out = subprocess.run([exe, "100"], capture_output=True, text=True, timeout=300).stdout.splitlines()
    g = load_golden("ladder_X25519.json")
    assert out[2] == g["kat"][0]["out"]
    assert out[out.index("chain 100") + 1] == g["ref_main_chain"]["checkpoints"]["100"]
    assert out[out.index("Alice shared secret") + 1] == g["ref_main_chain"]["dh"]["shared"]
    assert out[out.index("Bob's shared secret") + 1] == g["ref_main_chain"]["dh"]["shared"]
    assert out[-1] == "batched: equal"


def test_c_field_batch_example(torch_cuda, tmp_path):
    """examples/field_batch_tiles.c: element-major host arrays -> tiles on the device -> the generators' acceptance chain batched
    (modinv in place) -> back, from plain C through the C-ABI, checked there against the scalar entry points"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "field_batch_tiles")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "field_batch_tiles.c"), "-I", os.path.join(root, "include"),
                           "-L", os.path.join(root, "modarith_amd"), "-l:libmodarith_amd.so",
                           "-Wl,-rpath," + os.path.join(root, "modarith_amd"), "-o", exe])
    for n in ("21717", "70001"):                         # a partial last tile; a batch large enough for the shared inversions
        p = subprocess.run([exe, n], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "equal to the scalar entry points" in p.stdout, p.stdout[-500:] + p.stderr[-500:]


@pytest.mark.parametrize("P", EXTRA)
def test_extra_primes_vs_generic_oracle(oracle, torch_cuda, P):
    """the further primes: seeded batches against the run-time generic oracle (oracle/field_generic.c, itself
    pinned to the reference's golden vectors by tests/test_generic_oracle.py)"""
    from modarith_amd.field import Field
    from tests.util import derive_any as derive
    from tests.generic_oracle import Generic
    G = Generic(oracle.lib, P)
    F = Field(P)
    fp = derive(P)
    n = 4099
    rng = np.random.default_rng(77)
    def rnd(seed_off):
        out = rng.integers(0, 1 << fp.radix, size=(fp.nlimbs, n), dtype=np.uint64)
        out[fp.nlimbs - 1] = rng.integers(0, 1 << (fp.n - fp.radix * (fp.nlimbs - 1)), size=n, dtype=np.uint64)
        return np.ascontiguousarray(out)
    a, b = rnd(0), rnd(1)
    da, db = to_dev(a), to_dev(b)
    def gen(op, x, y=None):
        c = np.empty_like(x)
        G.lib.gen_batch(G.R, op, vp(x), vp(y) if y is not None else None, vp(c), n, n)
        return c
    assert np.array_equal(to_np(F.modmul(da, db)), gen(0, a, b))
    assert np.array_equal(to_np(F.modadd(da, db)), gen(1, a, b))
    assert np.array_equal(to_np(F.modsub(da, db)), gen(2, a, b))
    assert np.array_equal(to_np(F.modsqr(da)), gen(3, a))
    assert np.array_equal(to_np(F.modneg(da)), gen(4, a))
    assert np.array_equal(to_np(F.nres(da)), gen(5, a))
    assert np.array_equal(to_np(F.redc(da)), gen(6, a))
    for k in (3, 121665, 0x7fffffff):
        c = np.empty_like(a)
        G.lib.gen_batch_mli(G.R, vp(a), k, vp(c), n, n)
        assert np.array_equal(to_np(F.modmli(da, k)), c)
    m = 257
    sa = np.ascontiguousarray(a[:, :m])
    ci, cs = np.empty_like(sa), np.empty_like(sa)
    G.lib.gen_batch(G.R, 7, vp(sa), None, vp(ci), m, m)
    G.lib.gen_batch(G.R, 8, vp(sa), None, vp(cs), m, m)
    red = lambda x: (lambda c: (G.lib.gen_batch(G.R, 6, vp(x), None, vp(c), m, m), c)[1])(np.empty_like(x))
    dsa = to_dev(sa)
    assert np.array_equal(to_np(F.redc(F.modinv(dsa))), red(ci))
    assert np.array_equal(to_np(F.redc(F.modsqrt(dsa))), red(cs))


@pytest.mark.parametrize("P", ["X25519", "X448", "NIST224", "NIST384", "NIST521"])
def test_aos_soa_converters(torch_cuda, P):
    """element-major <-> limb-interleaved on the device, for 4/5/7/8/9-limb fields, incl. a padded limb stride"""
    torch = torch_cuda
    from modarith_amd.field import Field
    F = Field(P)
    n = 1003
    aos = torch.randint(0, 1 << 62, (n, F.N), dtype=torch.int64, device="cuda")
    soa = F.from_aos(aos)
    assert soa.shape == (F.N, n) and torch.equal(soa, aos.t())
    assert torch.equal(F.to_aos(soa), aos)
    wide = torch.zeros((F.N, n + 13), dtype=torch.int64, device="cuda")
    wide[:, :n] = soa
    assert torch.equal(F.to_aos(wide[:, :n]), aos)          # ld = n + 13
    assert F.from_aos(aos[:0]).shape == (F.N, 0)


def test_error_reporting(torch_cuda):
    """the reference signals no errors; the shim reports only misuse it cannot execute: misaligned byte records,
    a too-small ecn workspace.  The message is retrievable and the call leaves no sticky device error."""
    import torch
    from modarith_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    out = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    rc = lib.rfc7748_X25519_batch(buf.data_ptr() + 1, buf.data_ptr(), out.data_ptr(), 4, None)
    assert rc != 0 and b"aligned" in lib.modarith_amd_last_error()
    P = torch.zeros((3, 5, 8), dtype=torch.int64, device="cuda")
    rc = lib.ecn_ed25519_mul_batch(buf.data_ptr(), P.data_ptr(), 8, 8, None, 0, None)
    assert rc != 0 and b"workspace" in lib.modarith_amd_last_error()
    with pytest.raises(_lib.DeviceError):
        _lib.check(rc, "ecn_ed25519_mul_batch")
    # a correct call afterwards works
    from modarith_amd.field import Field
    F = Field("X25519")
    assert F.modis0(F.modzer(3)).cpu().tolist() == [1, 1, 1]


def test_stream_capture_into_hip_graph(torch_cuda):
    """the batched entry points only enqueue kernels on the caller's stream (no allocation, no synchronisation), so a
    sequence of them can be captured into a hipGraph and replayed on new data: launch-bound small batches pay one
    graph launch instead of one launch per call."""
    torch = torch_cuda
    from modarith_amd.field import Field, rfc7748
    F = Field("X25519")
    n = 512
    g = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda", generator=g)
    b = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda", generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    u = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    t1, t2, t3, out = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a), torch.empty_like(u)

    def body():
        F.modmul(a, b, out=t1)
        F.modsqr(t1, out=t2)
        F.modadd(t2, a, out=t3)
        F.modinv(t3, out=t1)
        rfc7748("X25519", k, u, out=out)

    body()                                   # eager reference (also loads the code objects before capture)
    torch.cuda.synchronize()
    want = (t1.clone(), t3.clone(), out.clone())
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            body()
    for buf in (t1, t2, t3, out):
        buf.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(t1, want[0]) and torch.equal(t3, want[1]) and torch.equal(out, want[2])
    # new inputs in the captured buffers, replay again
    a.copy_(b)
    u.copy_(k)
    graph.replay()
    torch.cuda.synchronize()
    got = (t1.clone(), t3.clone(), out.clone())
    body()
    torch.cuda.synchronize()
    assert torch.equal(t1, got[0]) and torch.equal(t3, got[1]) and torch.equal(out, got[2])


@pytest.mark.parametrize("P,fn", [("X25519", "modmul"), ("X448", "modsqr"), ("NIST256", "modadd")])
def test_host_pipeline_through_c_abi(oracle, torch_cuda, P, fn):
    """host-resident batch, chunked upload / kernel / download on three streams through the exported utilities only
    (modarith_amd/hostio.py), ragged last chunk, against the oracle"""
    from modarith_amd.hostio import PinnedArray, map_host
    N = PRIMES[P][0]
    n = 3 * 4096 + 1234
    a, b, c = PinnedArray(N, n), PinnedArray(N, n), PinnedArray(N, n)
    a.array[:] = random_soa(P, n, 91)
    b.array[:] = random_soa(P, n, 92)
    c.array[:] = 0
    map_host(P, fn, a, b, c, chunk=4096)
    ha, hb = np.ascontiguousarray(a.array), np.ascontiguousarray(b.array)
    want = oracle_bin(oracle, fn, P, ha, hb) if fn in ("modmul", "modadd") else oracle_un(oracle, fn, P, ha)
    assert np.array_equal(c.array, want)
    for x in (a, b, c):
        x.close()
