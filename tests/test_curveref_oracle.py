"""The oracle's curve layer (oracle/edwards_body.inc, weierstrass_body.inc) against the projective limbs the REFERENCE'S OWN
edwards.c / weierstrass.c produce (tests/golden/curveref_<CURVE>.json, made in the build container by
tests/golden/make_curveref.py from the reference's files: no model in between).  Inputs are raw limbs, outputs are compared limb
for limb: mul, dbl, add, sub, neg, cof, mul2, isinf, the generator, and the special cases.  CPU only."""
import ctypes

import pytest

from tests.conftest import load_golden

CURVES = ["ED25519", "ED448", "NUMS256E", "ED248", "ED376", "ED500", "NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W"]
SMALL_X = {"NUMS256E", "ED248", "ED376", "ED500", "NUMS256W"}          # ecnXXXgen takes a square root there: not in the fixture


def point(Pt, rows):
    p = Pt()
    for c, row in zip("xyz", rows):
        for i, v in enumerate(row):
            getattr(p, c)[i] = int(v, 16)
    return p


def rows(p):
    return [[hex(v) for v in getattr(p, c)] for c in "xyz"]


@pytest.fixture(scope="module", params=CURVES)
def cx(request, oracle):
    name = request.param
    C = name.lower()
    return name, C, oracle, oracle.ed[C][0], load_golden("curveref_%s.json" % name)


def test_generator_limbs(cx):
    name, C, o, Pt, g = cx
    p = Pt()
    o.ecn(C, "gen")(ctypes.byref(p))
    if name not in SMALL_X:
        assert rows(p) == g["gen"]
    else:
        # the reference's ecnXXXgen recovers y with modsqrt there, whose limbs depend on the external addition chain (SURVEY 8c caveat
        # 1): the fixture holds nres of the canonical coordinates instead, and the comparison is by value
        q = point(Pt, g["gen"])
        assert o.ecn(C, "cmp")(ctypes.byref(p), ctypes.byref(q)) == 1


def test_records_limb_for_limb(cx):
    name, C, o, Pt, g = cx
    ref = ctypes.byref
    cp = lambda p: Pt.from_buffer_copy(bytes(p))
    f = lambda fn: o.ecn(C, fn)
    for k, r in enumerate(g["records"]):
        e, fb = bytes.fromhex(r["e"]), bytes.fromhex(r["f"])
        M = point(Pt, r["P"]); f("mul")(e, ref(M)); assert rows(M) == r["M"], (name, k, "mul")
        D = cp(M); f("dbl")(ref(D)); assert rows(D) == r["D"], (name, k, "dbl")
        A = cp(M); f("add")(ref(D), ref(A)); assert rows(A) == r["A"], (name, k, "add")
        S = cp(A); f("sub")(ref(D), ref(S)); assert rows(S) == r["S"], (name, k, "sub")
        N = cp(A); f("neg")(ref(N)); assert rows(N) == r["N"], (name, k, "neg")
        Cf = cp(A); f("cof")(ref(Cf)); assert rows(Cf) == r["C"], (name, k, "cof")
        R = Pt(); m2, d2 = cp(M), cp(D); f("mul2")(e, ref(m2), fb, ref(d2), ref(R)); assert rows(R) == r["R"], (name, k, "mul2")
        Z = cp(A); f("add")(ref(N), ref(Z)); assert rows(Z) == r["A+N"] and f("isinf")(ref(Z)) == r["A+N_isinf"], (name, k, "P + (-P)")
        T, T2 = cp(A), cp(A); f("add")(ref(T2), ref(T)); assert rows(T) == r["A+A"], (name, k, "P + P through add")
        assert [f("isinf")(ref(x)) for x in (M, D, A, R)] == r["isinf"]


def test_special_cases(cx):
    name, C, o, Pt, g = cx
    ref = ctypes.byref
    f = lambda fn: o.ecn(C, fn)
    sp = g["special"]
    O = Pt(); f("inf")(ref(O)); assert rows(O) == sp["inf"]
    X = Pt.from_buffer_copy(bytes(O)); f("dbl")(ref(X)); assert rows(X) == sp["dbl_inf"]
    G = point(Pt, g["gen"])
    X = Pt.from_buffer_copy(bytes(G)); f("add")(ref(O), ref(X)); assert rows(X) == sp["gen+inf"]
    X = Pt.from_buffer_copy(bytes(O)); f("add")(ref(G), ref(X)); assert rows(X) == sp["inf+gen"]


def test_set_from_both_coordinates(cx):
    """ecnXXXset(s, x, y): big-endian coordinates in, limbs out (or the point at infinity for off-curve input)"""
    name, C, o, Pt, g = cx
    for r in g["set_xy"]:
        p = Pt()
        o.ecn(C, "set")(0, bytes.fromhex(r["x"]), bytes.fromhex(r["y"]), ctypes.byref(p))
        assert rows(p) == r["P"] and o.ecn(C, "isinf")(ctypes.byref(p)) == r["isinf"], (name, r["x"])
