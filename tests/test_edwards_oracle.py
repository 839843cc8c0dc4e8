"""The oracle's Edwards layer (restatement of edwards.c) against the big-integer fixtures
(tests/golden/edwards_*.json): scalar multiplication, add/dbl/sub, compression, mul2, the
testcurve.c checks and chain (whose 10000-step value equals the one captured from the reference run,
SURVEY 8 f1), RFC 8032 test 1.  CPU only."""
import ctypes

import pytest

from tests.conftest import load_golden

CURVES = [("ed25519", "ED25519"), ("ed448", "ED448"), ("nums256e", "NUMS256E"), ("ed248", "ED248"), ("ed376", "ED376"), ("ed500", "ED500")]


@pytest.fixture(scope="module", params=CURVES)
def cx(request, oracle):
    C, name = request.param
    return C, load_golden("edwards_%s.json" % name), oracle


def inf_xy(nb):
    return [(0).to_bytes(nb, "big").hex(), (1).to_bytes(nb, "big").hex()]


def test_generator(cx):
    C, g, o = cx
    Pt, nb = o.ed[C]
    p = Pt()
    o.ecn(C, "gen")(ctypes.byref(p))
    assert o.ed_xy(C, p) == g["gen"]
    assert not o.ecn(C, "isinf")(ctypes.byref(p))


def test_mul(cx):
    C, g, o = cx
    for rec in g["mul"]:
        p = o.ed_point(C, *rec["P"])
        o.ecn(C, "mul")(bytes.fromhex(rec["e"]), ctypes.byref(p))
        assert o.ed_xy(C, p) == rec["eP"], rec["e"]


def test_add_dbl_sub_cof_cmp(cx):
    C, g, o = cx
    for rec in g["ops"]:
        P, Q = o.ed_point(C, *rec["P"]), o.ed_point(C, *rec["Q"])
        o.ecn(C, "add")(ctypes.byref(Q), ctypes.byref(P))
        assert o.ed_xy(C, P) == rec["P+Q"]
        P = o.ed_point(C, *rec["P"])
        o.ecn(C, "dbl")(ctypes.byref(P))
        assert o.ed_xy(C, P) == rec["2P"]
        P2 = o.ed_point(C, *rec["2P"])
        assert o.ecn(C, "cmp")(ctypes.byref(P), ctypes.byref(P2)) == 1      # projective vs affine form
        P = o.ed_point(C, *rec["P"])
        o.ecn(C, "sub")(ctypes.byref(Q), ctypes.byref(P))
        assert o.ed_xy(C, P) == rec["P-Q"]
        P = o.ed_point(C, *rec["P"])
        o.ecn(C, "cof")(ctypes.byref(P))
        assert o.ed_xy(C, P) == rec["cofP"]
        assert o.ecn(C, "cmp")(ctypes.byref(P), ctypes.byref(Q)) == (1 if rec["cofP"] == rec["Q"] else 0)


def test_compress_decompress(cx):
    C, g, o = cx
    Pt, nb = o.ed[C]
    for rec in g["compress"]:
        p = Pt()
        o.ecn(C, "set")(rec["sy"], bytes.fromhex(rec["x"]), None, ctypes.byref(p))       # x + sign of y
        if rec["valid"]:
            assert o.ed_xy(C, p) == [rec["x"], rec["y"]]
            x = ctypes.create_string_buffer(nb)
            assert o.ecn(C, "get")(ctypes.byref(p), x, None) == rec["sy"] and x.raw.hex() == rec["x"]
        else:
            assert o.ecn(C, "isinf")(ctypes.byref(p))
        p = Pt()
        o.ecn(C, "set")(rec["sx"], None, bytes.fromhex(rec["y"]), ctypes.byref(p))       # y + sign of x
        if rec["valid"]:
            assert o.ed_xy(C, p) == [rec["x"], rec["y"]]
            y = ctypes.create_string_buffer(nb)
            assert o.ecn(C, "get")(ctypes.byref(p), None, y) == rec["sx"] and y.raw.hex() == rec["y"]
        else:
            assert o.ecn(C, "isinf")(ctypes.byref(p))
    for rec in g["set_xy"]:
        p = o.ed_point(C, rec["x"], rec["y"])
        assert bool(o.ecn(C, "isinf")(ctypes.byref(p))) == (not rec["valid"])


def test_mul2(cx):
    C, g, o = cx
    Pt, nb = o.ed[C]
    for rec in g["mul2"]:
        P, Q, R = o.ed_point(C, *rec["P"]), o.ed_point(C, *rec["Q"]), Pt()
        o.ecn(C, "mul2")(bytes.fromhex(rec["e"]), ctypes.byref(P), bytes.fromhex(rec["f"]), ctypes.byref(Q), ctypes.byref(R))
        assert o.ed_xy(C, R) == rec["R"]


@pytest.mark.parametrize("C,name", [("ed25519", "ED25519"), ("nums256e", "NUMS256E"), ("ed248", "ED248")])
def test_testcurve_checks_and_chain(oracle, C, name):
    """testcurve.c:224-255: order*G = O, r1*G + r2*G = O, then P = n1*P chained; 1000 steps here (for ED25519 the
    10000-step value in the fixture equals the reference's own output)."""
    o = oracle
    g = load_golden("edwards_%s.json" % name)
    t = g["testcurve"]
    Pt, nb = o.ed[C]
    P, Q = Pt(), Pt()
    o.ecn(C, "gen")(ctypes.byref(P))
    o.ecn(C, "cpy")(ctypes.byref(P), ctypes.byref(Q))
    o.ecn(C, "mul")(bytes.fromhex(t["order"]), ctypes.byref(P))
    assert o.ecn(C, "isinf")(ctypes.byref(P))
    o.ecn(C, "mul2")(bytes.fromhex(t["r1"]), ctypes.byref(Q), bytes.fromhex(t["r2"]), ctypes.byref(Q), ctypes.byref(P))
    assert o.ecn(C, "isinf")(ctypes.byref(P))
    o.ecn(C, "cpy")(ctypes.byref(Q), ctypes.byref(P))
    n1 = bytes.fromhex(t["n1"])
    for i in range(1000):
        o.ecn(C, "mul")(n1, ctypes.byref(P))
        if str(i + 1) in t["mul_chain"]:
            assert o.ed_xy(C, P) == t["mul_chain"][str(i + 1)]
    if name == "ED25519":
      assert t["mul_chain"]["10000"] == ["2c9de69f607e8732f75af34dd730c375c1df45dfebf036671fd483d6fd716c7d",
                                         "00cb089602e82a83c5952ac8d9b7ce1cad70696c97b220d0c514ea374cabe28d"]  # SURVEY 8(f1) probe


def test_rfc8032_public_key(oracle):
    o, C = oracle, "ed25519"
    g = load_golden("edwards_ED25519.json")["rfc8032_test1"]
    Pt, nb = o.ed[C]
    P = Pt()
    o.ecn(C, "gen")(ctypes.byref(P))
    o.ecn(C, "mul")(bytes.fromhex(g["scalar_be"]), ctypes.byref(P))
    y = ctypes.create_string_buffer(nb)
    sx = o.ecn(C, "get")(ctypes.byref(P), None, y)
    enc = bytearray(y.raw[::-1])
    enc[31] |= sx << 7
    assert enc.hex() == g["pk"]
