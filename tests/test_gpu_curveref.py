"""The batched curve kernels (csrc/curve.h, edwards.h, weierstrass.h) against the projective limbs the REFERENCE'S OWN
edwards.c / weierstrass.c produce (tests/golden/curveref_<CURVE>.json; tests/golden/make_curveref.py): raw limbs in, raw limbs
out, limb for limb -- mul, dbl, add, sub, neg, cof, mul2 in its exact form, isinf, the generator, the special cases; the default
(constant-time) mul2 by value (see there) -- for all eleven curves and the two generated ones.
Every record of a fixture is one lane of one batch, so the kernels run them side by side."""
import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
CURVES = ["ED25519", "ED448", "NUMS256E", "ED248", "ED376", "ED500", "NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W",
          # generated curves (modarith_amd.generate.EXAMPLE_CURVES; fixtures from the reference's code given the same definitions)
          "CURVE1174", "NIST224"]
SMALL_X = {"NUMS256E", "ED248", "ED376", "ED500", "NUMS256W"}


@pytest.fixture(scope="module", params=CURVES)
def cx(request):
    import torch
    assert torch.cuda.is_available()
    from modarith_amd.edwards import Curve
    name = request.param
    return name, Curve(name), load_golden("curveref_%s.json" % name), torch


def batch(torch, points):
    """list of [[x limbs], [y limbs], [z limbs]] (hex) -> int64 [3, N, n]"""
    a = np.array([[[int(v, 16) for v in row] for row in p] for p in points], dtype=np.uint64)          # [n, 3, N]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(1, 2, 0)).view(np.int64)).cuda()


def unbatch(t):
    a = t.cpu().numpy().view(np.uint64)
    return [[[hex(int(v)) for v in a[c, :, j]] for c in range(3)] for j in range(a.shape[2])]


def scalars(torch, hexes):
    return torch.tensor([list(bytes.fromhex(h)) for h in hexes], dtype=torch.uint8, device="cuda")


def test_generator_limbs(cx):
    name, W, g, torch = cx
    G = W.gen(3)
    if name not in SMALL_X:
        assert unbatch(G) == [g["gen"]] * 3
    else:                                  # ecnXXXgen takes a square root there (limbs depend on the external addition chain): by value
        assert W.cmp(G, batch(torch, [g["gen"]] * 3)).cpu().tolist() == [1, 1, 1]


def test_records_limb_for_limb(cx):
    name, W, g, torch = cx
    R = g["records"]
    col = lambda k: [r[k] for r in R]
    e, f = scalars(torch, col("e")), scalars(torch, col("f"))
    M = W.mul(e, batch(torch, col("P")))
    assert unbatch(M) == col("M"), "mul"
    D = W.dbl(M.clone())
    assert unbatch(D) == col("D"), "dbl"
    A = W.add(D, M.clone())
    assert unbatch(A) == col("A"), "add"
    assert unbatch(W.sub(D, A.clone())) == col("S"), "sub"
    N = W.neg(A.clone())
    assert unbatch(N) == col("N"), "neg"
    assert unbatch(W.cof(A.clone())) == col("C"), "cof"
    # mul2: the reference walks a joint sparse form with data-dependent branches ("not constant time", edwards.c:404-431, 486-510); the
    # kernel runs two fixed-window multiplications that share their doublings -- constant time, no lane divergence -- and reaches the
    # same POINT in another projective representative (csrc/curve.h mul2; include/modarith_amd.h says so).  Compared by value with
    # the reference's limbs (ecnXXXcmp: cross-multiplication, no inversion); the oracle reproduces the reference's limbs themselves
    assert unbatch(W.mul2(e, M.clone(), f, D.clone(), exact=True)) == col("R"), "mul2 (the reference's own walk: its limbs)"
    R2 = W.mul2(e, M.clone(), f, D.clone())
    assert W.cmp(R2, batch(torch, col("R"))).cpu().tolist() == [1] * len(R), "mul2"
    Z = W.add(N, A.clone())
    assert unbatch(Z) == col("A+N") and W.isinf(Z).cpu().tolist() == col("A+N_isinf"), "P + (-P)"
    assert unbatch(W.add(A.clone(), A.clone())) == col("A+A"), "P + P through add"
    flags = [W.isinf(x).cpu().tolist() for x in (M, D, A, R2)]
    assert [list(t) for t in zip(*flags)] == col("isinf")


def test_special_cases(cx):
    name, W, g, torch = cx
    sp = g["special"]
    O = W.inf(2)
    assert unbatch(O) == [sp["inf"]] * 2
    assert unbatch(W.dbl(O.clone())) == [sp["dbl_inf"]] * 2
    G = batch(torch, [g["gen"]] * 2)
    assert unbatch(W.add(O, G.clone())) == [sp["gen+inf"]] * 2
    assert unbatch(W.add(G, O.clone())) == [sp["inf+gen"]] * 2


def test_set_from_both_coordinates(cx):
    """ecnXXXset(s, x, y): big-endian coordinates in, limbs out (or the point at infinity for off-curve input)"""
    name, W, g, torch = cx
    recs = g["set_xy"]
    P = W.set(None, scalars(torch, [r["x"] for r in recs]), scalars(torch, [r["y"] for r in recs]))
    assert unbatch(P) == [r["P"] for r in recs]
    assert W.isinf(P).cpu().tolist() == [r["isinf"] for r in recs]


@pytest.mark.parametrize("name", ["CURVE1174", "NIST224"])
def test_generated_curve_affine_results(name):
    """a curve that is not in curve.py's table, end to end: key generation e -> affine e*G through gen / mul / get, a double
    multiplication through mul2 / get, compression and decompression through get / set -- against plain integer arithmetic on
    the curve's equation (the projective limbs are covered by the reference-run fixture above)"""
    import random
    import torch
    from modarith_amd import generate as gen
    from modarith_amd.edwards import Curve
    c = next(x for x in gen.EXAMPLE_CURVES if x["name"] == name)
    W = Curve(name)
    from modarith_amd.generate import params_of_plugin
    from modarith_amd.params import NAMED, derive
    p = (derive(c["field"]) if c["field"] in NAMED else params_of_plugin(c["field"])).p
    a, b, G = c["a"], c["b"], (c["gx"], c["gy"])
    if c["kind"] == "edwards":
        O = (0, 1)
        def add(P, Q):
            t = b * P[0] * Q[0] * P[1] * Q[1] % p
            return ((P[0] * Q[1] + P[1] * Q[0]) * pow(1 + t, -1, p) % p, (P[1] * Q[1] - a * P[0] * Q[0]) * pow(1 - t, -1, p) % p)
    else:
        O = None
        def add(P, Q):
            if P is None: return Q
            if Q is None: return P
            if P[0] == Q[0] and (P[1] + Q[1]) % p == 0: return None
            m = ((3 * P[0] * P[0] + a) * pow(2 * P[1], -1, p) if P == Q else (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p)) % p
            x = (m * m - P[0] - Q[0]) % p
            return (x, (m * (P[0] - x) - P[1]) % p)
    def mul(k, P):
        R = O
        while k:
            if k & 1: R = add(R, P)
            P = add(P, P); k >>= 1
        return R
    rng = random.Random(41)
    n, nb = 33, W.nbytes
    es = [rng.randrange(1, c["order"]) for _ in range(n)]
    fs = [rng.randrange(1, c["order"]) for _ in range(n)]
    es[0], es[1] = 1, c["order"] - 1
    rec = lambda ks: torch.tensor([list(k.to_bytes(nb, "big")) for k in ks], dtype=torch.uint8, device="cuda")
    def affine(Pt):
        x, y, _ = W.get(Pt.clone())
        return [(int.from_bytes(bytes(u), "big"), int.from_bytes(bytes(v), "big")) for u, v in zip(x.cpu().numpy(), y.cpu().numpy())]
    E = W.mul(rec(es), W.gen(n))
    want = [mul(e, G) for e in es]
    assert affine(E) == want
    Q = W.mul(rec(fs), W.gen(n))                                   # second base points
    R = W.mul2(rec(es), E.clone(), rec(fs), Q.clone())               # e*(eG) + f*(fG)
    assert affine(R) == [mul((e * e + f * f) % c["order"], G) for e, f in zip(es, fs)]
    assert W.isinf(W.mul(rec([c["order"]] * 2), W.gen(2))).cpu().tolist() == [1, 1]
    # compression: x and the sign of y  ->  the same point
    xb, yb, _ = W.get(E.clone())
    sgn = torch.tensor([w[1] & 1 for w in want], dtype=torch.int32, device="cuda")
    back = W.set(sgn, xb, None)
    assert affine(back) == want


@pytest.mark.parametrize("name", ["ED25519", "ED448", "NIST256", "NIST521"])
def test_exact_mul2_against_the_oracle_on_mixed_waves(oracle, name):
    """mul2(exact=True) on lanes whose walks differ -- full-size scalars next to tiny ones (1, 2, 3, 2^k), whose joint sparse forms
    start hundreds of digits later -- against the oracle's mul2, which reproduces the reference's limbs (tests/test_curveref_oracle.py)"""
    import ctypes
    import random
    import torch
    from modarith_amd.edwards import Curve
    W = Curve(name)
    C = name.lower()
    Pt, nb = oracle.ed[C]
    rng = random.Random(77)
    n = 192
    es = [rng.randrange(1, 1 << (8 * nb - 8)) for _ in range(n)]
    fs = [rng.randrange(1, 1 << (8 * nb - 8)) for _ in range(n)]
    for j, v in enumerate((1, 2, 3, 5, 1 << 20, (1 << 64) - 1)):
        es[7 * j + 1] = v
        fs[7 * j + 2] = v
        es[7 * j + 3], fs[7 * j + 3] = v, v + 1
    es[100], fs[100] = 1, 0
    es[101], fs[101] = 0, 1
    rec = lambda ks: torch.tensor([list(k.to_bytes(nb, "big")) for k in ks], dtype=torch.uint8, device="cuda")
    k0 = rec([rng.randrange(1, 1 << 200) for _ in range(n)])
    P = W.mul(k0, W.gen(n))
    Q = W.dbl(W.mul(rec([rng.randrange(1, 1 << 200) for _ in range(n)]), W.gen(n)))
    R = W.mul2(rec(es), P.clone(), rec(fs), Q.clone(), exact=True)
    got = R.cpu().numpy().view(np.uint64)
    hp, hq = P.cpu().numpy().view(np.uint64), Q.cpu().numpy().view(np.uint64)
    for j in range(n):
        p, q, r = Pt(), Pt(), Pt()
        for c, nm in enumerate("xyz"):
            for i in range(W.N):
                getattr(p, nm)[i] = int(hp[c, i, j]); getattr(q, nm)[i] = int(hq[c, i, j])
        oracle.ecn(C, "mul2")(es[j].to_bytes(nb, "big"), ctypes.byref(p), fs[j].to_bytes(nb, "big"), ctypes.byref(q), ctypes.byref(r))
        for c, nm in enumerate("xyz"):
            assert [int(v) for v in got[c, :, j]] == list(getattr(r, nm)), (name, j, nm)
    # and the default form reaches the same points
    assert W.cmp(W.mul2(rec(es), P.clone(), rec(fs), Q.clone()), R).cpu().tolist() == [1] * n


@pytest.mark.parametrize("name", ["ED25519", "ED448", "NIST256", "SECP256K1", "NUMS256E", "NIST521", "ED248", "ED376", "ED500", "NIST384", "NUMS256W"])
def test_points_beyond_the_limb_budget_return_the_references_limbs(oracle, name):
    """round 6 (csrc/curve.h "the limb contract"): the reference's ecnXXXmul over the pasted field.c returns defined limbs for EVERY 64-bit
    limb pattern (edwards.c:435-482); the kernels' fast classes (FieldH51 / FieldH56, the FAST products) are exact only inside the budget
    2^(Radix+2).  A wave that holds a point beyond it is left to a second launch on the exact class.  Four kinds of waves side by side:
    ordinary points; ordinary points with ONE lane whose limbs run up to 2^63; fat but valid representatives (the same field elements with
    2^(Radix+3) moved from limb 1 into limb 0); random 64-bit limbs in every lane.  mul and mul2(exact) limb for limb against the oracle
    (which is the reference's field.c for all inputs: tests/test_oracle_golden.py, 64-bit limb classes); the default mul2 by value
    on the valid representatives."""
    import ctypes
    import random
    import torch
    from modarith_amd.edwards import Curve
    W = Curve(name)
    C = name.lower()
    Pt, nb = oracle.ed[C]
    from modarith_amd import curves
    from modarith_amd.params import derive
    N = W.N
    radix = derive((curves.CURVES[name] if name in curves.CURVES else curves.W_CURVES[name]).field).radix
    rng = random.Random(606)
    n = 6 * 64
    rec = lambda ks: torch.tensor([list(k.to_bytes(nb, "big")) for k in ks], dtype=torch.uint8, device="cuda")
    es = [rng.randrange(1, 1 << (8 * nb - 8)) for _ in range(n)]
    fs = [rng.randrange(1, 1 << (8 * nb - 8)) for _ in range(n)]
    P = W.mul(rec([rng.randrange(1, 1 << 200) for _ in range(n)]), W.gen(n))
    Q = W.dbl(W.mul(rec([rng.randrange(1, 1 << 200) for _ in range(n)]), W.gen(n)))
    hp, hq = P.cpu().numpy().view(np.uint64).copy(), Q.cpu().numpy().view(np.uint64).copy()
    M64 = (1 << 64) - 1
    hp[1, 2, 64 + 17] = (1 << 63) + 12345                                  # wave 1: one lane, one limb
    hq[0, 0, 128 + 5] = (1 << 62) + 99                                     # wave 2: one lane of Q
    fat = slice(192, 256)                                                  # wave 3: fat but valid representatives
    for h in (hp, hq):
        for c in range(3):
            t = np.uint64(1 << 3)
            ok = h[c, 1, fat] >= t
            h[c, 0, fat] += np.where(ok, np.uint64(1 << (radix + 3)), np.uint64(0))
            h[c, 1, fat] -= np.where(ok, t, np.uint64(0))
    for j in range(256, 320):                                              # wave 4: random 64-bit limbs everywhere
        for c in range(3):
            for i in range(N):
                hp[c, i, j] = rng.getrandbits(64) & M64
                hq[c, i, j] = rng.getrandbits(64 if j % 2 else radix + 2) & M64
    Pd, Qd = torch.from_numpy(hp.view(np.int64)).cuda(), torch.from_numpy(hq.view(np.int64)).cuda()
    assert not bool(W.limbs_ok(Pd)[64 + 17]) and bool(W.limbs_ok(Pd)[0]) and not bool(W.limbs_ok(Pd)[200])
    Mg = W.mul(rec(es), Pd.clone()).cpu().numpy().view(np.uint64)
    Xg = W.mul2(rec(es), Pd.clone(), rec(fs), Qd.clone(), exact=True).cpu().numpy().view(np.uint64)
    Dg = W.mul2(rec(es), Pd.clone(), rec(fs), Qd.clone())

    def pt(h, j):
        p = Pt()
        for c, nm in enumerate("xyz"):
            for i in range(N):
                getattr(p, nm)[i] = int(h[c, i, j])
        return p

    want2 = np.zeros_like(Xg)
    for j in range(n):
        p, q, r = pt(hp, j), pt(hq, j), Pt()
        oracle.ecn(C, "mul2")(es[j].to_bytes(nb, "big"), ctypes.byref(p), fs[j].to_bytes(nb, "big"), ctypes.byref(q), ctypes.byref(r))
        oracle.ecn(C, "mul")(es[j].to_bytes(nb, "big"), ctypes.byref(p))
        for c, nm in enumerate("xyz"):
            assert [int(v) for v in Mg[c, :, j]] == list(getattr(p, nm)), (name, "mul", j, nm)
            assert [int(v) for v in Xg[c, :, j]] == list(getattr(r, nm)), (name, "mul2 exact", j, nm)
            want2[c, :, j] = list(getattr(r, nm))
    valid = list(range(0, 64)) + list(range(192, 256)) + list(range(320, n))            # lanes whose points ARE points of the curve
    same = W.cmp(Dg, torch.from_numpy(want2.view(np.int64)).cuda()).cpu().tolist()
    assert [same[j] for j in valid] == [1] * len(valid)


@pytest.mark.parametrize("name", ["ED25519", "NIST256", "ED448"])
def test_scalar_entry_points_return_the_references_limbs(name):
    """the scalar API (host pointers, curve.h's own signatures): mul, dbl, add and mul2 on the fixture's records, limb for limb --
    the scalar mul2 takes the reference's own walk (one element: nothing to keep in step)"""
    import ctypes
    from modarith_amd import _lib
    lib = _lib.load()
    g = load_golden("curveref_%s.json" % name)
    N, nb, c = g["N"], g["Nbytes"], name.lower()

    class Pt(ctypes.Structure):
        _fields_ = [("x", ctypes.c_uint64 * N), ("y", ctypes.c_uint64 * N), ("z", ctypes.c_uint64 * N)]
    def point(rows):
        p = Pt()
        for k, row in zip("xyz", rows):
            for i, v in enumerate(row):
                getattr(p, k)[i] = int(v, 16)
        return p
    rows = lambda p: [[hex(v) for v in getattr(p, k)] for k in "xyz"]
    f = lambda fn: getattr(lib, "ecn_%s_%s" % (c, fn))
    ref = ctypes.byref
    for r in g["records"][:4]:
        e, fb = bytes.fromhex(r["e"]), bytes.fromhex(r["f"])
        M = point(r["P"]); f("mul")(e, ref(M)); assert rows(M) == r["M"]
        D = point(r["M"]); f("dbl")(ref(D)); assert rows(D) == r["D"]
        A = point(r["M"]); f("add")(ref(D), ref(A)); assert rows(A) == r["A"]
        R = Pt(); f("mul2")(e, ref(M), fb, ref(D), ref(R)); assert rows(R) == r["R"], "scalar mul2"
