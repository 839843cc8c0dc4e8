"""The oracle's Weierstrass layer (restatement of weierstrass.c; NIST P-256 on the per-prime field restatement, NIST
P-384 on the generic field oracle bound to the reference's constants) against the big-integer fixtures
(tests/golden/weierstrass_<CURVE>.json) and the testcurve.c checks.  CPU only."""
import ctypes

import pytest

from tests.conftest import load_golden

C = "nist256"


@pytest.fixture(autouse=True, params=["nist256", "nist384", "nist521", "secp256k1", "nums256w"])
def _curve(request):
    global C
    C = request.param


def gold():
    return load_golden("weierstrass_%s.json" % C.upper())


def pt(o, xy):
    return o.ed_point(C, xy[0], xy[1])


def test_generator_and_mul(oracle):
    o, g = oracle, gold()
    Pt, nb = o.ed[C]
    p = Pt()
    o.ecn(C, "gen")(ctypes.byref(p))
    assert o.ed_xy(C, p) == g["gen"]
    for rec in g["mul"]:
        p = pt(o, rec["P"])
        o.ecn(C, "mul")(bytes.fromhex(rec["e"]), ctypes.byref(p))
        assert bool(o.ecn(C, "isinf")(ctypes.byref(p))) == bool(rec["inf"])
        assert o.ed_xy(C, p) == rec["eP"], rec["e"]


def test_add_dbl_sub(oracle):
    o, g = oracle, gold()
    for rec in g["ops"]:
        P, Q = pt(o, rec["P"]), pt(o, rec["Q"])
        o.ecn(C, "add")(ctypes.byref(Q), ctypes.byref(P))          # complete: also P+P and P+(-P)
        assert o.ed_xy(C, P) == rec["P+Q"] and bool(o.ecn(C, "isinf")(ctypes.byref(P))) == bool(rec["P+Q_inf"])
        P = pt(o, rec["P"])
        o.ecn(C, "dbl")(ctypes.byref(P))
        assert o.ed_xy(C, P) == rec["2P"]
        P2 = pt(o, rec["2P"])
        assert o.ecn(C, "cmp")(ctypes.byref(P), ctypes.byref(P2)) == 1
        P = pt(o, rec["P"])
        o.ecn(C, "sub")(ctypes.byref(Q), ctypes.byref(P))
        assert o.ed_xy(C, P) == rec["P-Q"] and bool(o.ecn(C, "isinf")(ctypes.byref(P))) == bool(rec["P-Q_inf"])


def test_decompress_and_set(oracle):
    o, g = oracle, gold()
    Pt, nb = o.ed[C]
    for rec in g["compress"]:
        p = Pt()
        o.ecn(C, "set")(rec["sy"], bytes.fromhex(rec["x"]), None, ctypes.byref(p))
        if rec["valid"]:
            assert o.ed_xy(C, p) == [rec["x"], rec["y"]]
            x = ctypes.create_string_buffer(nb)
            assert o.ecn(C, "get")(ctypes.byref(p), x, None) == rec["sy"] and x.raw.hex() == rec["x"]
        else:
            assert o.ecn(C, "isinf")(ctypes.byref(p))
    for rec in g["set_xy"]:
        p = pt(o, [rec["x"], rec["y"]])
        assert bool(o.ecn(C, "isinf")(ctypes.byref(p))) == (not rec["valid"])


def test_mul2_and_testcurve(oracle):
    o, g = oracle, gold()
    Pt, nb = o.ed[C]
    for rec in g["mul2"]:
        P, Q, R = pt(o, rec["P"]), pt(o, rec["Q"]), Pt()
        o.ecn(C, "mul2")(bytes.fromhex(rec["e"]), ctypes.byref(P), bytes.fromhex(rec["f"]), ctypes.byref(Q), ctypes.byref(R))
        assert o.ed_xy(C, R) == rec["R"]
    t = g["testcurve"]
    P, Q = Pt(), Pt()
    o.ecn(C, "gen")(ctypes.byref(P))
    o.ecn(C, "cpy")(ctypes.byref(P), ctypes.byref(Q))
    o.ecn(C, "mul")(bytes.fromhex(t["order"]), ctypes.byref(P))
    assert o.ecn(C, "isinf")(ctypes.byref(P))                                   # testcurve.c:224-229
    o.ecn(C, "mul2")(bytes.fromhex(t["r1"]), ctypes.byref(Q), bytes.fromhex(t["r2"]), ctypes.byref(Q), ctypes.byref(P))
    assert o.ecn(C, "isinf")(ctypes.byref(P))                                   # testcurve.c:231-237
    o.ecn(C, "cpy")(ctypes.byref(Q), ctypes.byref(P))
    n1 = bytes.fromhex(t["n1"])
    for i in range(1000):                                                       # testcurve.c:247-255
        o.ecn(C, "mul")(n1, ctypes.byref(P))
        if str(i + 1) in t["mul_chain"]:
            assert o.ed_xy(C, P) == t["mul_chain"][str(i + 1)]
