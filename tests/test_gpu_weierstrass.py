"""GPU parity of the batched Weierstrass layer (NIST P-256, P-384 and P-521): HIP kernels through the C-ABI
against the big-integer fixture (affine, canonical) and against the oracle's restatement of weierstrass.c
limb for limb (projective coordinates) for add, dbl, mul."""
import ctypes

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
C = "nist256"


@pytest.fixture(scope="module", params=["NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W"])
def cx(request):
    global C
    import torch
    assert torch.cuda.is_available()
    from modarith_amd.edwards import Curve
    C = request.param.lower()
    return Curve(request.param), load_golden("weierstrass_%s.json" % request.param), torch


def dev_bytes(torch, hexes):
    return torch.tensor([list(bytes.fromhex(h)) for h in hexes], dtype=torch.uint8, device="cuda")


def points(W, torch, xy_list):
    return W.set(None, dev_bytes(torch, [p[0] for p in xy_list]), dev_bytes(torch, [p[1] for p in xy_list]))


def xy_of(W, P):
    x, y, _ = W.get(P.clone())
    return [[bytes(a).hex(), bytes(b).hex()] for a, b in zip(x.cpu().numpy(), y.cpu().numpy())]


def test_gen_mul(cx):
    W, g, torch = cx
    assert xy_of(W, W.gen(2)) == [g["gen"]] * 2
    assert W.isinf(W.inf(2)).cpu().tolist() == [1, 1]
    recs = g["mul"]
    P = points(W, torch, [r["P"] for r in recs])
    W.mul(dev_bytes(torch, [r["e"] for r in recs]), P)
    assert W.isinf(P).cpu().tolist() == [r["inf"] for r in recs]
    assert xy_of(W, P) == [r["eP"] for r in recs]


def test_ops(cx):
    W, g, torch = cx
    recs = g["ops"]
    P0 = points(W, torch, [r["P"] for r in recs])
    Q = points(W, torch, [r["Q"] for r in recs])
    S = W.add(Q, P0.clone())
    assert xy_of(W, S) == [r["P+Q"] for r in recs] and W.isinf(S).cpu().tolist() == [r["P+Q_inf"] for r in recs]
    assert xy_of(W, W.dbl(P0.clone())) == [r["2P"] for r in recs]
    D = W.sub(Q, P0.clone())
    assert xy_of(W, D) == [r["P-Q"] for r in recs] and W.isinf(D).cpu().tolist() == [r["P-Q_inf"] for r in recs]
    two = W.dbl(P0.clone())
    assert W.cmp(two, points(W, torch, [r["2P"] for r in recs])).cpu().tolist() == [1] * len(recs)
    assert W.cmp(W.ran(5, two.clone()), two).cpu().tolist() == [1] * len(recs)


def test_decompress_set_mul2(cx):
    W, g, torch = cx
    recs = g["compress"]
    sy = torch.tensor([r["sy"] for r in recs], dtype=torch.int32, device="cuda")
    P = W.set(sy, dev_bytes(torch, [r["x"] for r in recs]), None)
    valid = [r["valid"] for r in recs]
    assert W.isinf(P).cpu().tolist() == [1 - v for v in valid]
    for r, xy in zip(recs, xy_of(W, P)):
        if r["valid"]:
            assert xy == [r["x"], r["y"]]
    x, y, sign = W.get(P.clone(), want_x=True, want_y=False)
    assert [s for s, v in zip(sign.cpu().tolist(), valid) if v] == [r["sy"] for r in recs if r["valid"]]
    sxy = g["set_xy"]
    assert W.isinf(points(W, torch, [[r["x"], r["y"]] for r in sxy])).cpu().tolist() == [1 - r["valid"] for r in sxy]
    with pytest.raises(Exception):
        W.set(None, None, dev_bytes(torch, [recs[0]["y"]]))          # weierstrass.c needs x
    m2 = g["mul2"]
    R = W.mul2(dev_bytes(torch, [r["e"] for r in m2]), points(W, torch, [r["P"] for r in m2]),
               dev_bytes(torch, [r["f"] for r in m2]), points(W, torch, [r["Q"] for r in m2]))
    assert xy_of(W, R) == [r["R"] for r in m2]


def test_testcurve(cx):
    W, g, torch = cx
    t = g["testcurve"]
    lanes = 66
    G = W.gen(lanes)
    assert W.isinf(W.mul(dev_bytes(torch, [t["order"]] * lanes), G.clone())).cpu().tolist() == [1] * lanes
    R = W.mul2(dev_bytes(torch, [t["r1"]] * lanes), G, dev_bytes(torch, [t["r2"]] * lanes), G)
    assert W.isinf(R).cpu().tolist() == [1] * lanes
    n1 = dev_bytes(torch, [t["n1"]] * lanes)
    P = G.clone()
    for i in range(100):
        W.mul(n1, P)
        if str(i + 1) in t["mul_chain"]:
            assert xy_of(W, P) == [t["mul_chain"][str(i + 1)]] * lanes


def test_projective_limbs_equal_oracle(oracle, cx):
    W, g, torch = cx
    Pt, nb = oracle.ed[C]
    nl, n = W.N, 1200
    rng = np.random.default_rng(23)
    k0 = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    e = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    base = W.mul(torch.from_numpy(k0).cuda(), W.gen(n))
    soa = np.ascontiguousarray(base.cpu().numpy().view(np.uint64))
    want = soa.reshape(3 * nl, n).copy()
    oracle.ecn(C, "batch_mul")(e.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), n, n)
    got = W.mul(torch.from_numpy(e).cuda(), base.clone()).cpu().numpy().view(np.uint64).reshape(3 * nl, n)
    assert np.array_equal(got, want)
    m = 48
    Q = W.dbl(base[:, :, :m].contiguous().clone())
    S = W.add(Q, base[:, :, :m].contiguous().clone())
    qn, sn = Q.cpu().numpy().view(np.uint64), S.cpu().numpy().view(np.uint64)
    for j in range(m):
        p = Pt()
        for c, nm in enumerate(("x", "y", "z")):
            for i in range(nl):
                getattr(p, nm)[i] = int(soa[c, i, j])
        q = Pt()
        oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(q))
        oracle.ecn(C, "dbl")(ctypes.byref(q))
        oracle.ecn(C, "add")(ctypes.byref(q), ctypes.byref(p))
        for c, nm in enumerate(("x", "y", "z")):
            assert [int(v) for v in qn[c, :, j]] == list(getattr(q, nm))
            assert [int(v) for v in sn[c, :, j]] == list(getattr(p, nm))


def test_c_curve_api_example(tmp_path):
    """examples/ecn_batch.c: curve.h's API under the reference's names in plain C -- testcurve.c's order / r1+r2 checks
    through the scalar entry points, then a batched key generation compared with the scalar path."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ecn_batch")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "ecn_batch.c"), "-I", os.path.join(root, "include"),
                           "-L", os.path.join(root, "modarith_amd"), "-l:libmodarith_amd.so",
                           "-Wl,-rpath," + os.path.join(root, "modarith_amd"), "-o", exe])
    p = subprocess.run([exe, "2048"], capture_output=True, text=True, timeout=300)
    out = p.stdout.splitlines()
    assert p.returncode == 0, p.stdout + p.stderr
    assert out[1] == "79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798"
    assert out[2].endswith("yes") and out[3].endswith("yes")
    assert out[-1] == "batched == scalar: equal" and out[-2] == "fused == call sequences: equal"
