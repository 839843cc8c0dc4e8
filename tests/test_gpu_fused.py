"""GPU parity of the fused scalar multiplication + affine export (csrc/ed26.h, ed28.h, wn26.h; ecn_<c>_mul_get_batch): byte-equal to
ecn_<c>_mul_batch followed by ecn_<c>_get_batch, to the reference-derived fixtures, and to the CPU oracle's ecn mul +
ecn get -- on random projective points, the special points of the curve and corner scalars."""
import ctypes

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
FUSED = [("ed25519", "ED25519"), ("ed448", "ED448"), ("nist256", "NIST256"), ("secp256k1", "SECP256K1")]
WEIER = ("nist256", "secp256k1")


@pytest.fixture(scope="module", params=FUSED)
def fx(request):
    import torch
    assert torch.cuda.is_available()
    from modarith_amd.edwards import Edwards
    C, name = request.param
    return C, Edwards(name), load_golden(("weierstrass_%s.json" if C in WEIER else "edwards_%s.json") % name), torch


def dev_bytes(torch, hexes):
    return torch.tensor([list(bytes.fromhex(h)) for h in hexes], dtype=torch.uint8, device="cuda")


def hexrows(t):
    return [bytes(r).hex() for r in t.cpu().numpy()]


def test_fused_mul_get_fixture(fx):
    """the reference-derived records of edwards_<C>.json (tests/golden/make_golden.py): e, P -> affine e*P"""
    C, Ed, g, torch = fx
    recs = g["mul"]
    P = Ed.set(None, dev_bytes(torch, [r["P"][0] for r in recs]), dev_bytes(torch, [r["P"][1] for r in recs]))
    keep = P.clone()
    x, y, sign = Ed.mul_get(dev_bytes(torch, [r["e"] for r in recs]), P)
    assert torch.equal(P, keep)                                              # the point batch is not modified
    assert [[a, b] for a, b in zip(hexrows(x), hexrows(y))] == [r["eP"] for r in recs]
    assert sign.cpu().tolist() == [0] * len(recs)                            # both coordinates requested
    # the reference main()'s chain P = n1*P (testcurve.c:247-255) through the fused kernel: steps 1 and 10
    if "testcurve" in g:
        t = g["testcurve"]
        n1 = dev_bytes(torch, [t["n1"]])
        Pp = Ed.gen(1)
        for i in range(10):
            x, y, _ = Ed.mul_get(n1, Pp)
            if str(i + 1) in t["mul_chain"]:
                assert [hexrows(x)[0], hexrows(y)[0]] == t["mul_chain"][str(i + 1)]
            Pp = Ed.set(None, x, y)


def test_fused_equals_two_call_form_random(fx, oracle):
    """2^16 random (scalar, projective point) pairs: fused == mul + get on the GPU == the oracle's mul + get"""
    C, Ed, g, torch = fx
    n = 1 << (16 if C == "ed25519" else 14)
    gen = torch.Generator(device="cuda").manual_seed(91)
    k0 = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    e = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    base = Ed.mul(k0, Ed.gen(n))                                             # random points, projective (Z != 1)
    x, y, _ = Ed.mul_get(e, base)
    want = Ed.mul(e, base.clone())
    wx, wy, _ = Ed.get(want)
    assert torch.equal(x, wx) and torch.equal(y, wy)
    # the oracle on a sample (scalar functions, element by element)
    Pt, nb = oracle.ed[C]
    soa = base.cpu().numpy().view(np.uint64)
    he, hx, hy = e.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy()
    for j in list(range(0, n, 997)) + [n - 1]:
        p = Pt()
        for c, nm in enumerate(("x", "y", "z")):
            for i in range(Ed.N):
                getattr(p, nm)[i] = int(soa[c, i, j])
        oracle.ecn(C, "mul")(bytes(he[j]), ctypes.byref(p))
        ox, oy = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        oracle.ecn(C, "get")(ctypes.byref(p), ox, oy)
        assert bytes(hx[j]) == ox.raw and bytes(hy[j]) == oy.raw, j
    # one coordinate + sign (point compression, edwards.c:219-239)
    xo, none, sy = Ed.mul_get(e[:4099], base[:, :, :4099].contiguous(), want_y=False)
    assert none is None and torch.equal(xo, x[:4099]) and torch.equal(sy, (y[:4099, -1] & 1).to(torch.int32))
    none, yo, sx = Ed.mul_get(e[:4099], base[:, :, :4099].contiguous(), want_x=False)
    assert none is None and torch.equal(yo, y[:4099]) and torch.equal(sx, (x[:4099, -1] & 1).to(torch.int32))


def test_fused_special_points_and_scalars(fx):
    """the complete addition law at work: neutral element, points of order 2, 4 and 8, scalars 0, 1, 8, the group order,
    order +- 1 and all ones -- every (point, scalar) pair against the two-call form"""
    C, Ed, g, torch = fx
    if C in WEIER:
        pytest.skip("Edwards special points; see test_fused_weierstrass_special_points_and_scalars")
    p = (1 << 255) - 19 if C == "ed25519" else (1 << 448) - (1 << 224) - 1
    order = int(g["order"], 16)
    be = lambda v: v.to_bytes(Ed.nbytes, "big").hex()
    G = Ed.gen(1)
    lowy = [0, p - 1, 1]                                                     # y = 0 (order 4), y = -1 (order 2), y = 1 (neutral)
    pts = [G, Ed.inf(1)]
    for yv in lowy:
        for s in (0, 1):
            pts.append(Ed.set(torch.tensor([s], dtype=torch.int32, device="cuda"), None, dev_bytes(torch, [be(yv)])))
    # a point of order 8: (order * k) * (random point of the full group); decompress y values until one has order 8
    for yv in range(2, 40):
        Q = Ed.set(torch.tensor([0], dtype=torch.int32, device="cuda"), None, dev_bytes(torch, [be(yv)]))
        if Ed.isinf(Q).item():
            continue
        T = Ed.mul(dev_bytes(torch, [be(order)]), Q.clone())                 # kills the prime-order part
        if not Ed.isinf(Ed.mul(dev_bytes(torch, [be(4)]), T.clone())).item():
            pts.append(T)                                                    # 4T != O: order 8
            break
    assert len(pts) >= (9 if C == "ed25519" else 8)          # (ED448 has cofactor 4: no point of order 8)
    nbits = 8 * Ed.nbytes
    scalars = [0, 1, 2, 7, 8, order - 1, order, order + 1, 4 * order - 1, (1 << nbits) - 1, 1 << (nbits - 1), (1 << (nbits - 1)) - 1]
    P = torch.cat([q for q in pts for _ in scalars], dim=2).contiguous()
    e = dev_bytes(torch, [be(s) for _ in pts for s in scalars])
    x, y, _ = Ed.mul_get(e, P)
    wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    # order * G is the neutral element: affine (0, 1)
    i = scalars.index(order)
    assert hexrows(x)[i] == be(0) and hexrows(y)[i] == be(1)


def test_fused_weierstrass_special_points_and_scalars(fx):
    """prime-order curve, complete formulas: the point at infinity, G, -G, a random point and its negative against scalars
    0, 1, 2, 8, 15, 16, the group order q, q +- 1, 2q - 1, all ones, single high bits -- against the two-call form; an
    infinite result leaves as (0, 1), what ecnXXXget gives (weierstrass.c:299-310)"""
    C, Ed, g, torch = fx
    if C not in WEIER:
        pytest.skip("Weierstrass curves")
    order = int(g["order"], 16)
    be = lambda v: v.to_bytes(Ed.nbytes, "big").hex()
    gen = torch.Generator(device="cuda").manual_seed(5)
    G = Ed.gen(1)
    Rn = Ed.mul(torch.randint(0, 256, (1, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=gen), Ed.gen(1))
    pts = [G, Ed.inf(1), Ed.neg(G.clone()), Rn, Ed.neg(Rn.clone())]
    nbits = 8 * Ed.nbytes
    scalars = [0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88, order - 1, order, order + 1, (1 << nbits) - 1, 1 << (nbits - 1), (1 << (nbits - 1)) - 1,
               (1 << nbits) - order, 0x8888888888888888888888888888888888888888888888888888888888888888 % (1 << nbits), (1 << nbits) - 8]
    P = torch.cat([q for q in pts for _ in scalars], dim=2).contiguous()
    e = dev_bytes(torch, [be(s) for _ in pts for s in scalars])
    x, y, _ = Ed.mul_get(e, P)
    wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    i = scalars.index(order)
    assert hexrows(x)[i] == be(0) and hexrows(y)[i] == be(1)                 # q * G = infinity -> (0, 1)
    # mul2: e*P + f*Q = infinity (Q = -P, f = e), P = Q, infinite operands
    n = len(scalars)
    Pm = torch.cat([Rn] * n, dim=2).contiguous()
    Qm = torch.cat([Ed.neg(Rn.clone())] * n, dim=2).contiguous()
    em = dev_bytes(torch, [be(s) for s in scalars])
    x2, y2, _ = Ed.mul2_get(em, Pm, em, Qm)
    assert hexrows(x2) == [be(0)] * n and hexrows(y2) == [be(1)] * n
    w2x, w2y, _ = Ed.get(Ed.mul2(em, Pm, em, Qm))
    assert torch.equal(x2, w2x) and torch.equal(y2, w2y)


def test_fused_weierstrass_scalars_of_the_exceptional_cases(fx):
    """round 5: P-256 runs k P in Jacobian coordinates with the exceptional cases decided by the scalar (csrc/wj26.h), secp256k1 splits
    the scalar by the endomorphism (csrc/glv26.h).  The scalars where those forms have something to get wrong -- q - 2m and q + 2m (the
    Jacobian accumulator meets +-Q at the last digit), leading zero windows, single digits, multiples of 16, lambda, q - lambda, 2^128
    +- 1, values between q and 2^256 -- on G, -G, a random point, its negative and the point at infinity: mul_get against ecn mul +
    ecn get (the reference's complete formulas, bit-exact to the oracle in tests/test_gpu_weierstrass.py), mulgen2_get against
    ecn mul2 + ecn get with the generator as first point, every scalar of the list as f and as e"""
    C, Ed, g, torch = fx
    if C not in WEIER:
        pytest.skip("Weierstrass curves")
    q = int(g["order"], 16)
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    be = lambda v: v.to_bytes(Ed.nbytes, "big").hex()
    gen = torch.Generator(device="cuda").manual_seed(55)
    G = Ed.gen(1)
    Rn = Ed.mul(torch.randint(0, 256, (1, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=gen), Ed.gen(1))
    pts = [G, Ed.neg(G.clone()), Rn, Ed.neg(Rn.clone()), Ed.inf(1)]
    scalars = [q - 2 * m for m in range(1, 9)] + [q + 2 * m for m in range(1, 9)] + [q - m for m in (1, 3, 15, 16, 17)] + list(range(0, 18))
    scalars += [d << (4 * i) for i in (1, 2, 31, 32, 33, 62, 63) for d in (1, 7, 8, 9, 15)]
    scalars += [lam, q - lam, lam + 1, lam - 1, (lam * 2) % q, 2**128, 2**128 - 1, 2**128 + 1, 2**129, 2**256 - 1, 2**256 - 2, 2**256 - q, 2 * q - 2**256 + 5, q // 2, (q + 1) // 2]
    P = torch.cat([p for p in pts for _ in scalars], dim=2).contiguous()
    e = dev_bytes(torch, [be(s) for _ in pts for s in scalars])
    x, y, _ = Ed.mul_get(e, P)
    wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    # e G on the fixed-base table (P-256: Jacobian mixed additions): the list above, top comb digit 0 / 1 / 2, e = d 2^256 mod q, single comb digits
    gs = scalars + [2**256 - q, 2 * (2**256 - q), (2**256 - q) + 2**255, q - 2**255, q - 2**255 + 1, 3 * 2**254 - 1, 3 * 2**254, 3 * 2**254 + 1]
    gs += [d << (5 * i) for i in range(0, 52) for d in (1, 15, 16, 17, 31) if (d << (5 * i)) < 2**256]
    eg = dev_bytes(torch, [be(v) for v in gs])
    gx, gy, _ = Ed.mulgen_get(eg)
    wgx, wgy, _ = Ed.get(Ed.mul(eg, Ed.gen(len(gs))))
    assert torch.equal(gx, wgx) and torch.equal(gy, wgy)
    n = e.shape[0]
    other = dev_bytes(torch, [be(scalars[(7 * i + 3) % len(scalars)]) for i in range(n)])
    for ee, ff in ((other, e), (e, other), (e, e)):
        x2, y2, _ = Ed.mulgen2_get(ee, ff, P)
        w2x, w2y, _ = Ed.get(Ed.mul2(ee, Ed.gen(n), ff, P.clone()))
        assert torch.equal(x2, w2x) and torch.equal(y2, w2y)
        x3, y3, _ = Ed.mul2_get(ee, Ed.gen(n), ff, P)
        assert torch.equal(x3, w2x) and torch.equal(y3, w2y)


def test_fused_more_points_than_resident_lanes(fx):
    """the kernels hold one table slot per RESIDENT lane (131 072) and walk larger batches grid-stride, rebuilding the
    table in the same slot: 2 full passes + a ragged third one, against the two-call form"""
    C, Ed, g, torch = fx
    n = 2 * 131072 + 77
    gen = torch.Generator(device="cuda").manual_seed(93)
    rnd = lambda m: torch.randint(0, 256, (m, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    P = Ed.mul(rnd(n), Ed.gen(n))
    e = rnd(n)
    x, y, _ = Ed.mul_get(e, P)
    wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    m = 131072 + 4099
    Pm, Qm, em, fm = P[:, :, :m].contiguous(), P[:, :, n - m:].contiguous(), e[:m].contiguous(), rnd(m)
    x, y, _ = Ed.mul2_get(em, Pm, fm, Qm)
    wx, wy, _ = Ed.get(Ed.mul2(em, Pm, fm, Qm))
    assert torch.equal(x, wx) and torch.equal(y, wy)


def test_fused_ladder_and_straus_forms_cross_their_chunk_boundary():
    """round 5: the ladder / Straus forms of the Edwards curves work through a batch in chunks of 2^20 records (csrc/edlad_k.h EDLAD_CHUNK:
    the workspace stops growing there) -- one full chunk and a ragged second one, with the exceptional inputs of the ladder form placed
    on both sides of the boundary (the neutral element, the point of order two, scalar 0, the group order), against the call-by-call
    forms; and the same call with NO workspace argument (the library's scratch pool), as callers of rounds 2-4 made it for ed25519"""
    import torch
    from modarith_amd.edwards import Curve
    from modarith_amd import _lib
    Ed = Curve("ED25519")
    n = (1 << 20) + 4099
    gen = torch.Generator(device="cuda").manual_seed(95)
    rnd = lambda m: torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda", generator=gen)
    e, f = rnd(n), rnd(n)
    P = Ed.mul(rnd(n), Ed.gen(n))
    q = bytes.fromhex("1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3ed")
    for j in (5, (1 << 20) - 1, 1 << 20, n - 3):
        P[:, :, j] = Ed.inf(1)[:, :, 0]                                  # neutral element on both sides of the chunk boundary
        e[j + 1] = 0                                                       # scalar 0
        e[j + 2] = torch.tensor(list(q), dtype=torch.uint8)                # [q]P = neutral
    two = Ed.set(torch.zeros(1, dtype=torch.int32, device="cuda"), None, torch.tensor([list(bytes.fromhex("7f" + "ff" * 30 + "ec"))], dtype=torch.uint8, device="cuda"))
    P[:, :, 9] = two[:, :, 0]                                            # (0, -1)
    P[:, :, (1 << 20) + 9] = two[:, :, 0]
    x, y, _ = Ed.mul_get(e, P)
    wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    x, y, _ = Ed.mulgen2_get(e, f, P)
    wx, wy, _ = Ed.get(Ed.mul2(e, Ed.gen(n), f, P))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    Q = Ed.mul(f, Ed.gen(n))
    x, y, _ = Ed.mul2_get(e, P, f, Q)
    wx, wy, _ = Ed.get(Ed.mul2(e, P, f, Q))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    # no workspace argument at all: the rounds-2..4 calling convention of ed25519 (workspace_bytes = 0, NULL)
    L = _lib.load()
    x2 = torch.empty_like(x); y2 = torch.empty_like(y)
    m = 8192 + 3
    _lib.check(L.ecn_ed25519_mul_get_batch(e.data_ptr(), P.data_ptr(), x2.data_ptr(), y2.data_ptr(), None, m, n, None, 0, None), "mul_get without a workspace")
    torch.cuda.synchronize()
    wx, wy, _ = Ed.get(Ed.mul(e[:m].contiguous(), P[:, :, :m].contiguous()))
    assert torch.equal(x2[:m], wx) and torch.equal(y2[:m], wy)
    del Ed, P, Q
    Ed = Curve("ED448")
    n = (1 << 20) + 130
    rnd = lambda m: torch.randint(0, 256, (m, 56), dtype=torch.uint8, device="cuda", generator=gen)
    e = rnd(n)
    P = Ed.mul(rnd(n), Ed.gen(n))
    P[:, :, (1 << 20)] = Ed.inf(1)[:, :, 0]
    x, y, _ = Ed.mul_get(e, P)
    wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)


def test_fused_p256_point_off_the_curve_leaves_its_neighbours_alone():
    """round-5 advisor (csrc/wn_affine.h): the affine window tables of P-256 share one inversion between the entries of up to four
    records of a lane's column.  A point off the curve with Y = 0 has Z = 0 in its entries 2P, 4P, 6P, 8P (Z3 = 2 Y Z) but not in entry
    P: the way up counted those entries as 1, the way down multiplied them in as 0 and zeroed the tables of every EARLIER record of the
    column -- in a verification batch one unvalidated public key spoiled other users' results.  Records off the curve placed second and
    third in their columns: every other record must equal the call-by-call form, in all three fused forms that build such tables."""
    import torch
    from modarith_amd.edwards import Curve
    W = Curve("NIST256")
    nb = W.nbytes
    n = (1 << 17) + 100                                     # one chunk; three records per column (ne = 8), two with two tables (ne = 16)
    gen = torch.Generator(device="cuda").manual_seed(1906)
    rnd = lambda m: torch.randint(0, 256, (m, nb), dtype=torch.uint8, device="cuda", generator=gen)
    e, f = rnd(n), rnd(n)
    P = W.mul(rnd(n), W.gen(n))
    Q = W.mul(rnd(n), W.gen(n))
    L3, L2 = (n + 2) // 3, (n + 1) // 2
    bad = sorted({L3 + 5, 2 * L3 + 7, L2 + 5, L2 + 11, n - 1})
    for j in bad:
        P[1, :, j] = 0                                      # (x, 0, Z): not on the curve (the group has odd order: no point has y = 0)
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[torch.tensor(bad, device="cuda")] = False
    x, y, _ = W.mul_get(e, P)
    wx, wy, _ = W.get(W.mul(e, P.clone()))
    assert torch.equal(x[keep], wx[keep]) and torch.equal(y[keep], wy[keep])
    assert int((x[keep] != 0).any(dim=1).sum()) > n - 10    # (and the neighbours are real points, not the zeros the bug left behind)
    x, y, _ = W.mulgen2_get(e, f, P)
    wx, wy, _ = W.get(W.mul2(e, W.gen(n), f, P))
    assert torch.equal(x[keep], wx[keep]) and torch.equal(y[keep], wy[keep])
    for A, B in ((P, Q), (Q, P)):                            # the bad points in the first table, then in the second
        x, y, _ = W.mul2_get(e, A, f, B)
        wx, wy, _ = W.get(W.mul2(e, A, f, B))
        assert torch.equal(x[keep], wx[keep]) and torch.equal(y[keep], wy[keep])


@pytest.mark.parametrize("name", ["NIST256", "SECP256K1"])
def test_fused_weierstrass_export_crosses_its_chunk_boundary(name):
    """round 5: the fused Weierstrass kernels hand (X : Y : Z) to an inversion shared by up to 32 records (csrc/wn_export.h) and work
    through a batch in chunks of 2^20 records: one full chunk and a ragged second one, points at infinity and scalars 0 / n (infinite
    results leave as (0, 1) and count as Z = 1 in the shared product) on both sides of the boundary and next to ordinary records of the
    same column, all three fused forms against the call-by-call forms; sign-only and x-only outputs"""
    import torch
    from modarith_amd.edwards import Curve
    W = Curve(name)
    nb = W.nbytes
    n = (1 << 20) + 2053
    gen = torch.Generator(device="cuda").manual_seed(97)
    rnd = lambda m: torch.randint(0, 256, (m, nb), dtype=torch.uint8, device="cuda", generator=gen)
    e, f = rnd(n), rnd(n)
    P = W.mul(rnd(n), W.gen(n))
    q = {"NIST256": "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551",
         "SECP256K1": "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141"}[name]
    qb = torch.tensor(list(bytes.fromhex(q)), dtype=torch.uint8)
    for j in (0, 7, 65536, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, n - 3):
        P[:, :, j] = W.inf(1)[:, :, 0]
        e[j + 1] = 0
        e[j + 2] = qb
    x, y, sg = W.mul_get(e, P)
    wx, wy, wsg = W.get(W.mul(e, P.clone()))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    z = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    one = z.clone(); one[-1] = 1
    for j in (0, 1, 2, 1 << 20, (1 << 20) + 1, (1 << 20) + 2):
        assert torch.equal(x[j], z) and torch.equal(y[j], one), j
    x, y, _ = W.mulgen2_get(e, f, P)
    wx, wy, _ = W.get(W.mul2(e, W.gen(n), f, P))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    m = (1 << 20) + 130
    Q = W.mul(f[:m].contiguous(), W.gen(m))
    Pm, em, fm = P[:, :, :m].contiguous(), e[:m].contiguous(), f[:m].contiguous()
    x, y, _ = W.mul2_get(em, Pm, fm, Q)
    wx, wy, _ = W.get(W.mul2(em, Pm, fm, Q))
    assert torch.equal(x, wx) and torch.equal(y, wy)


def test_fused_ladder_form_under_stream_capture():
    """the ladder form is four kernels and a workspace: with the caller's workspace it only enqueues kernels, so it can be captured into a
    hipGraph and replayed on new data; WITHOUT one it would have to take scratch from the library's pool, which a stream under capture
    does not allow -- the call then fails with hipErrorInvalidValue and a message, it does not break the capture or fall back"""
    import torch
    from modarith_amd.edwards import Curve
    from modarith_amd import _lib
    Ed = Curve("ED25519")
    n = 4096 + 5
    gen = torch.Generator(device="cuda").manual_seed(96)
    rnd = lambda: torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    e, e2 = rnd(), rnd()
    P = Ed.mul(rnd(), Ed.gen(n))
    want = Ed.mul_get(e, P)                                      # eager (also sizes the object's workspace before the capture)
    want2 = Ed.mul_get(e2, P)
    L = _lib.load()
    ws = torch.empty(int(L.ecn_ed25519_mul_get_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    x, y = torch.empty_like(want[0]), torch.empty_like(want[1])
    eb = e.clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc_none = L.ecn_ed25519_mul_get_batch(eb.data_ptr(), P.data_ptr(), x.data_ptr(), y.data_ptr(), None, n, n, None, 0, side.cuda_stream)
            msg = L.modarith_amd_last_error().decode()
            rc = L.ecn_ed25519_mul_get_batch(eb.data_ptr(), P.data_ptr(), x.data_ptr(), y.data_ptr(), None, n, n, ws.data_ptr(), ws.numel(), side.cuda_stream)
    assert rc == 0 and rc_none != 0 and "workspace" in msg, (rc, rc_none, msg)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, want[0]) and torch.equal(y, want[1])
    eb.copy_(e2)                                                 # new scalars in the captured buffer
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, want2[0]) and torch.equal(y, want2[1])


@pytest.mark.parametrize("c,name", [("ed25519", "ED25519"), ("ed448", "ED448")])
def test_straus_form_takes_a_workspace_at_any_address(c, name):
    """round-5 advisor: ecn_<c>_mul2_get_batch wanted its workspace 128-byte aligned and, handed a large enough one at another address,
    took the library's pool without a word -- or, under stream capture (no pool), failed with a message that named no reason.  The library
    now aligns inside the buffer (the reported size includes the slack).  Under capture, where only the caller's workspace can serve:
    addresses off by 8, 64 and 120 bytes work; a workspace short by one byte is refused and the message says it is too small."""
    import torch
    from modarith_amd.edwards import Curve
    from modarith_amd import _lib
    Ed = Curve(name)
    nb = Ed.nbytes
    n = 1000
    gen = torch.Generator(device="cuda").manual_seed(128)
    rnd = lambda: torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda", generator=gen)
    e, f = rnd(), rnd()
    P, Q = Ed.mul(rnd(), Ed.gen(n)), Ed.mul(rnd(), Ed.gen(n))
    want = Ed.mul2_get(e, P, f, Q)
    L = _lib.load()
    fn, szf = getattr(L, "ecn_%s_mul2_get_batch" % c), getattr(L, "ecn_%s_mul2_get_workspace_bytes" % c)
    need = int(szf(n))
    buf = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    base = (buf.data_ptr() + 127) // 128 * 128
    for off in (8, 64, 120, 0):
        x, y = torch.zeros_like(want[0]), torch.zeros_like(want[1])
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                rc = fn(e.data_ptr(), P.data_ptr(), f.data_ptr(), Q.data_ptr(), x.data_ptr(), y.data_ptr(), None, n, n, base + off, need, side.cuda_stream)
                rc_short = fn(e.data_ptr(), P.data_ptr(), f.data_ptr(), Q.data_ptr(), x.data_ptr(), y.data_ptr(), None, n, n, base + off, need - 128, side.cuda_stream)
                msg = L.modarith_amd_last_error().decode()
        assert rc == 0 and rc_short != 0 and "too small" in msg, (off, rc, rc_short, msg)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(x, want[0]) and torch.equal(y, want[1]), off


@pytest.mark.parametrize("name", ["NIST256", "SECP256K1"])
def test_fused_weierstrass_forms_under_stream_capture(name):
    """round 5: mul_get of the Weierstrass curves is two kernels per chunk on the caller's workspace (window kernel, shared inversion +
    export) and captures as it is; mulgen_get has no workspace argument and takes the shared inversion's scratch from the library's pool
    -- not available under capture, where it falls back to the four-scalars-per-lane kernel of rounds 2-4.  Both replayed on new data,
    against the eager calls"""
    import torch
    from modarith_amd.edwards import Curve
    from modarith_amd import _lib
    W = Curve(name)
    c = name.lower()
    n = 4096 + 5
    gen = torch.Generator(device="cuda").manual_seed(98)
    rnd = lambda: torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    e, e2 = rnd(), rnd()
    e[3] = 0
    P = W.mul(rnd(), W.gen(n))
    P[:, :, 7] = W.inf(1)[:, :, 0]
    want, want2 = W.mul_get(e, P), W.mul_get(e2, P)
    gwant, gwant2 = W.mulgen_get(e), W.mulgen_get(e2)             # eager: one scalar per lane through the scratch pool
    L = _lib.load()
    ws = torch.empty(int(getattr(L, "ecn_%s_mul_get_workspace_bytes" % c)(n)), dtype=torch.uint8, device="cuda")
    x, y, gx, gy = (torch.empty_like(want[0]) for _ in range(4))
    eb = e.clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = getattr(L, "ecn_%s_mul_get_batch" % c)(eb.data_ptr(), P.data_ptr(), x.data_ptr(), y.data_ptr(), None, n, n, ws.data_ptr(), ws.numel(), side.cuda_stream)
            rcg = getattr(L, "ecn_%s_mulgen_get_batch" % c)(eb.data_ptr(), gx.data_ptr(), gy.data_ptr(), None, n, side.cuda_stream)
    assert rc == 0 and rcg == 0, (rc, rcg, L.modarith_amd_last_error().decode())
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, want[0]) and torch.equal(y, want[1]) and torch.equal(gx, gwant[0]) and torch.equal(gy, gwant[1])
    eb.copy_(e2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, want2[0]) and torch.equal(y, want2[1]) and torch.equal(gx, gwant2[0]) and torch.equal(gy, gwant2[1])


def test_fused_rejects_bad_arguments(fx):
    C, Ed, g, torch = fx
    from modarith_amd.edwards import Edwards
    with pytest.raises(ValueError):
        Edwards("NUMS256W").mul_get(torch.zeros((1, 32), dtype=torch.uint8, device="cuda"), Edwards("NUMS256W").gen(1))
    x, y, s = Ed.mul_get(torch.zeros((0, Ed.nbytes), dtype=torch.uint8, device="cuda"), Ed.empty(0))
    assert x.shape[0] == 0
    if C in WEIER:
        # the Weierstrass calls live on the caller's workspace: missing, misaligned or short -> hipErrorInvalidValue and a message, nothing launched
        from modarith_amd import _lib
        L = _lib.load()
        n = 100
        e = torch.zeros((n, Ed.nbytes), dtype=torch.uint8, device="cuda")
        P = Ed.gen(n)
        xo, yo = torch.empty_like(e), torch.empty_like(e)
        need = int(getattr(L, "ecn_%s_mul_get_workspace_bytes" % C)(n))
        ws = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
        call = getattr(L, "ecn_%s_mul_get_batch" % C)
        for wp, wb in ((None, 0), (ws.data_ptr() + 1, need), (ws.data_ptr(), need - 8)):
            assert call(e.data_ptr(), P.data_ptr(), xo.data_ptr(), yo.data_ptr(), None, n, n, wp, wb, None) != 0
            assert "workspace" in L.modarith_amd_last_error().decode()
        assert call(e.data_ptr(), P.data_ptr(), xo.data_ptr(), yo.data_ptr(), None, n, n, ws.data_ptr(), need, None) == 0
        torch.cuda.synchronize()


@pytest.mark.parametrize("C,name,lg", [("ed25519", "ED25519", 14), ("ed448", "ED448", 12), ("nist256", "NIST256", 14), ("secp256k1", "SECP256K1", 14)])
def test_fused_mul2_get(oracle, C, name, lg):
    """e*P + f*Q and its affine export in one kernel: against mul2 + get on the GPU (2^14 / 2^12 random pairs) and the
    oracle's ecn mul2 + ecn get on a sample; special operands: neutral element, P = Q, zero scalars, small order"""
    import torch
    from modarith_amd.edwards import Edwards
    Ed = Edwards(name)
    n = 1 << lg
    gen = torch.Generator(device="cuda").manual_seed(92)
    rnd = lambda: torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    P, Q = Ed.mul(rnd(), Ed.gen(n)), Ed.mul(rnd(), Ed.gen(n))
    e, f = rnd(), rnd()
    P[:, :, 0:8] = Ed.inf(8)                        # neutral element as P
    Q[:, :, 8:16] = Ed.inf(8)                       # ... as Q
    Q[:, :, 16:24] = P[:, :, 16:24]                 # P = Q
    if C in WEIER:
        Q[:, :, 24:32] = Ed.neg(P[:, :, 24:32].contiguous())            # Q = -P
        f[24:28] = e[24:28]                                              # ... with f = e: the sum is the point at infinity
    else:
        low = Ed.set(torch.zeros(8, dtype=torch.int32, device="cuda"), None, torch.zeros((8, Ed.nbytes), dtype=torch.uint8, device="cuda"))
        Q[:, :, 24:32] = low                        # y = 0: order 4
    e[32:40] = 0
    f[40:48] = 0
    e[48:56] = 255
    f[48:56] = 255
    keepP, keepQ = P.clone(), Q.clone()
    x, y, _ = Ed.mul2_get(e, P, f, Q)
    assert torch.equal(P, keepP) and torch.equal(Q, keepQ)
    wx, wy, _ = Ed.get(Ed.mul2(e, P, f, Q))
    assert torch.equal(x, wx) and torch.equal(y, wy)
    Pt, nb = oracle.ed[C]
    sp, sq = P.cpu().numpy().view(np.uint64), Q.cpu().numpy().view(np.uint64)
    he, hf, hx, hy = e.cpu().numpy(), f.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy()
    for j in list(range(0, 64, 3)) + list(range(64, n, 1999 if lg > 12 else 499)):
        p, q, r = Pt(), Pt(), Pt()
        for c, nm in enumerate(("x", "y", "z")):
            for i in range(Ed.N):
                getattr(p, nm)[i] = int(sp[c, i, j])
                getattr(q, nm)[i] = int(sq[c, i, j])
        oracle.ecn(C, "mul2")(bytes(he[j]), ctypes.byref(p), bytes(hf[j]), ctypes.byref(q), ctypes.byref(r))
        ox, oy = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        oracle.ecn(C, "get")(ctypes.byref(r), ox, oy)
        assert bytes(hx[j]) == ox.raw and bytes(hy[j]) == oy.raw, j
    none, yo, sx = Ed.mul2_get(e[:100], P[:, :, :100].contiguous(), f[:100], Q[:, :, :100].contiguous(), want_x=False)
    assert none is None and torch.equal(yo, y[:100]) and torch.equal(sx, (x[:100, -1] & 1).to(torch.int32))


@pytest.mark.parametrize("name", ["ED25519", "ED448"])
def test_fused_generator_multiplication_edwards(oracle, name):
    """ecn gen + ecn mul + ecn get in one kernel on the Edwards curves: against the fused mul_get on the generator, the
    two-call form and, on a sample, the oracle; corner scalars; more scalars than resident lanes"""
    import torch
    from modarith_amd.edwards import Edwards
    W = Edwards(name)
    g = load_golden("edwards_%s.json" % name)
    order = int(g["order"], 16)
    nb = W.nbytes
    n = 4 * 65536 + 4099
    gen = torch.Generator(device="cuda").manual_seed(95)
    e = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda", generator=gen)
    top = 1 << (8 * nb)
    corner = [0, 1, 2, 8, 9, 15, 16, 0x88, order - 1, order, order + 1, (top - 1) // order * order, top - 1, top >> 1, int("8" * (2 * nb), 16), int("9" * (2 * nb), 16), int("7" * (2 * nb), 16)]
    e[:len(corner)] = dev_bytes(torch, [v.to_bytes(nb, "big").hex() for v in corner])
    x, y, s0 = W.mulgen_get(e)
    fx_, fy_, _ = W.mul_get(e, W.gen(n))
    assert torch.equal(x, fx_) and torch.equal(y, fy_)
    m = 1 << 13
    wx, wy, _ = W.get(W.mul(e[:m].contiguous(), W.gen(m)))
    assert torch.equal(x[:m], wx) and torch.equal(y[:m], wy)
    assert hexrows(x[:1]) == ["00" * nb] and hexrows(y[:1]) == ["00" * (nb - 1) + "01"]    # 0 * G = neutral element (0, 1)
    none, yo, sx = W.mulgen_get(e[:4099].contiguous(), want_x=False)                       # compressed public keys (ed448.c:181)
    assert none is None and torch.equal(yo, y[:4099]) and torch.equal(sx, (x[:4099, -1] & 1).to(torch.int32))
    C = name.lower()
    Pt, onb = oracle.ed[C]
    he, hx, hy = e.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy()
    for j in list(range(0, 20)) + list(range(20, n, 19991)):
        p = Pt()
        oracle.ecn(C, "gen")(ctypes.byref(p))
        oracle.ecn(C, "mul")(bytes(he[j]), ctypes.byref(p))
        ox, oy = ctypes.create_string_buffer(onb), ctypes.create_string_buffer(onb)
        oracle.ecn(C, "get")(ctypes.byref(p), ox, oy)
        assert bytes(hx[j]) == ox.raw and bytes(hy[j]) == oy.raw, j


@pytest.mark.parametrize("name", ["NIST256", "SECP256K1"])
def test_fused_generator_multiplication(oracle, name):
    """ecn gen + ecn mul + ecn get in one kernel (fixed-base table, mixed additions): against the fused mul_get on the
    generator, the two-call form and, on a sample, the oracle; corner scalars; more scalars than resident lanes"""
    import torch
    from modarith_amd.edwards import Edwards
    W = Edwards(name)
    g = load_golden("weierstrass_%s.json" % name)
    order = int(g["order"], 16)
    n = 4 * 65536 + 4099
    gen = torch.Generator(device="cuda").manual_seed(94)
    e = torch.randint(0, 256, (n, W.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    corner = [0, 1, 2, 8, 9, 15, 16, 0x88, order - 1, order, order + 1, (1 << 256) - 1, 1 << 255, (1 << 256) - order, int("8" * 64, 16), int("9" * 64, 16)]
    e[:len(corner)] = dev_bytes(torch, [v.to_bytes(32, "big").hex() for v in corner])
    x, y, s0 = W.mulgen_get(e)
    G = W.gen(n)
    fx_, fy_, _ = W.mul_get(e, G)
    assert torch.equal(x, fx_) and torch.equal(y, fy_)
    m = 1 << 14
    wx, wy, _ = W.get(W.mul(e[:m].contiguous(), W.gen(m)))
    assert torch.equal(x[:m], wx) and torch.equal(y[:m], wy)
    assert hexrows(x[:1]) == ["00" * 32] and hexrows(y[:1]) == ["00" * 31 + "01"]         # 0 * G = infinity -> (0, 1)
    assert [hexrows(x[1:2])[0], hexrows(y[1:2])[0]] == g["gen"]
    assert s0.cpu().tolist() == [0] * n
    xo, none, sy = W.mulgen_get(e[:4099].contiguous(), want_y=False)                       # compressed public keys
    assert none is None and torch.equal(xo, x[:4099]) and torch.equal(sy, (y[:4099, -1] & 1).to(torch.int32))
    C = name.lower()
    Pt, nb = oracle.ed[C]
    he, hx, hy = e.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy()
    for j in list(range(0, 20)) + list(range(20, n, 9973)):
        p = Pt()
        oracle.ecn(C, "gen")(ctypes.byref(p))
        oracle.ecn(C, "mul")(bytes(he[j]), ctypes.byref(p))
        ox, oy = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        oracle.ecn(C, "get")(ctypes.byref(p), ox, oy)
        assert bytes(hx[j]) == ox.raw and bytes(hy[j]) == oy.raw, j


@pytest.mark.parametrize("name", ["NIST256", "SECP256K1", "ED25519", "ED448"])
def test_fused_verification_form(oracle, name):
    """e*G + f*Q and its affine export (ecn gen + ecn mul2 + ecn get, the verification pattern): against the general fused
    mul2_get with P = G, the three-call form, and the oracle on a sample; Q infinite / +-G, zero scalars, f = e with Q = -G
    (infinite result), more pairs than resident lanes"""
    import torch
    from modarith_amd.edwards import Edwards
    W = Edwards(name)
    nb = W.nbytes
    n = 131072 + 8197
    gen = torch.Generator(device="cuda").manual_seed(96)
    rnd = lambda m: torch.randint(0, 256, (m, nb), dtype=torch.uint8, device="cuda", generator=gen)
    e, f = rnd(n), rnd(n)
    Q = W.mul(rnd(n), W.gen(n))
    Q[:, :, 0:8] = W.inf(8)
    Q[:, :, 8:16] = W.gen(8)
    Q[:, :, 16:24] = W.neg(W.gen(8))
    f[16:20] = e[16:20]                              # e G + e (-G): the neutral element / point at infinity -> (0, 1)
    e[24:28] = 0
    f[28:32] = 0
    e[32:34] = 255
    f[32:34] = 255
    keep = Q.clone()
    x, y, _ = W.mulgen2_get(e, f, Q)
    assert torch.equal(Q, keep)
    G = W.gen(n)
    gx_, gy_, _ = W.mul2_get(e, G, f, Q)
    assert torch.equal(x, gx_) and torch.equal(y, gy_)
    m = 1 << 13
    wx, wy, _ = W.get(W.mul2(e[:m].contiguous(), W.gen(m), f[:m].contiguous(), Q[:, :, :m].contiguous()))
    assert torch.equal(x[:m], wx) and torch.equal(y[:m], wy)
    assert hexrows(x[16:20]) == ["00" * nb] * 4 and hexrows(y[16:20]) == ["00" * (nb - 1) + "01"] * 4
    C = name.lower()
    Pt, onb = oracle.ed[C]
    sq = Q.cpu().numpy().view(np.uint64)
    he, hf, hx, hy = e.cpu().numpy(), f.cpu().numpy(), x.cpu().numpy(), y.cpu().numpy()
    for j in list(range(0, 36, 3)) + list(range(36, n, 14983)):
        g_, q, r = Pt(), Pt(), Pt()
        oracle.ecn(C, "gen")(ctypes.byref(g_))
        for c, nm in enumerate(("x", "y", "z")):
            for i in range(W.N):
                getattr(q, nm)[i] = int(sq[c, i, j])
        oracle.ecn(C, "mul2")(bytes(he[j]), ctypes.byref(g_), bytes(hf[j]), ctypes.byref(q), ctypes.byref(r))
        ox, oy = ctypes.create_string_buffer(onb), ctypes.create_string_buffer(onb)
        oracle.ecn(C, "get")(ctypes.byref(r), ox, oy)
        assert bytes(hx[j]) == ox.raw and bytes(hy[j]) == oy.raw, j


@pytest.mark.parametrize("curve", ["X25519", "X448"])
def test_rfc7748_on_the_base_point(oracle, curve):
    """rfc7748(k, base) from the fixed-base table of the equivalent Edwards curve: against the ladder kernel with u = 9 / 5 on
    2^18 + 77 random private keys and corner keys (zeros, ones, 0x88.., 0x77.., 4q for X448), and the oracle on a sample"""
    import torch
    from modarith_amd.field import rfc7748, rfc7748_base
    nb = 32 if curve == "X25519" else 56
    n = (1 << 18) + 77
    gen = torch.Generator(device="cuda").manual_seed(97)
    k = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda", generator=gen)
    k[0] = 0
    k[1] = 255
    k[2] = 0x88
    k[3] = 0x77
    if curve == "X448":
        q4 = 0xfffffffffffffffffffffffffffffffffffffffffffffffffffffffdf3288fa7113b6d26bb58da4085b309ca37163d548de30a4aad6113cc
        k[4] = torch.tensor(list(q4.to_bytes(56, "little")), dtype=torch.uint8, device="cuda")
    u = torch.zeros((n, nb), dtype=torch.uint8, device="cuda")
    u[:, 0] = 9 if curve == "X25519" else 5
    want = rfc7748(curve, k, u)
    got = rfc7748_base(curve, k)
    assert torch.equal(got, want)
    if curve == "X448":
        assert int(got[4].sum()) == 0                       # 4q * base = the point at infinity -> 0, as the ladder gives
    hk, hg, hu = k.cpu().numpy(), got.cpu().numpy(), bytes(u[0].cpu().numpy())
    for j in list(range(8)) + list(range(8, n, 33331)):
        assert bytes(hg[j]) == oracle.ladder(curve, bytes(hk[j]), hu), j
