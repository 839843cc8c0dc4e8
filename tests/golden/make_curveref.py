"""Projective-limb fixtures of the curve layer straight from the REFERENCE'S OWN edwards.c / weierstrass.c, built in this
container by tests/golden/curveref.py (no stand-ins; the functions that need the external addchain tool are simply not
called).  For every curve of curve.py's list: the generator's limbs, and chained records

    P (limbs, Z != 1 after the first) , e, f  ->  M = e*P (ecnXXXmul), D = 2M (dbl), A = M + D (add), S = A - D (sub),
    N = -A (neg), C = cof(A), R = e*M + f*D (mul2), isinf flags

plus the special cases P + P through add, P + (-P), multiplication by 0 and 1, operations on the point at infinity, and
ecnXXXset from both coordinates (on- and off-curve input).
Inputs and outputs are the struct's raw limbs, so the oracle's restatement and the HIP kernels are compared with the reference
limb for limb (tests/test_curveref_oracle.py, tests/test_gpu_curveref.py).   python tests/golden/make_curveref.py [CURVE ...]
"""
import ctypes, json, os, random, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import curveref  # noqa: E402
import gio  # noqa: E402

CURVES = ("ED25519", "ED448", "NUMS256E", "ED248", "ED376", "ED500", "NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W")


def custom_curves():
    """modarith_amd.generate.EXAMPLE_CURVES in curve.py's vocabulary"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from modarith_amd.generate import EXAMPLE_CURVES, resolve, EXAMPLES
    from modarith_amd.params import NAMED, derive
    out = {}
    for c in EXAMPLE_CURVES:
        if c["field"] in NAMED:
            fp, arg = derive(c["field"]), c["field"]
        else:
            arg, fam = next((a, f) for a, f in EXAMPLES if resolve(a, f).name == c["field"])
            fp, arg = resolve(arg, fam), arg.split("=", 1)[-1]
        out[c["name"]] = dict(p=fp.p, q=c["order"], cof=c.get("cof", 0), prime_type=fp.family, curve_type=c["kind"], A=c["a"], B=c["b"], X=c["gx"], Y=c["gy"], field_arg=arg)
    return out


def affine_multiples(cu, count):
    """[G, 2G, ...] as affine (x, y) by plain integer arithmetic, for the ecnXXXset inputs of a custom curve"""
    p, a, b = cu["p"], cu["A"], cu["B"]
    G = (cu["X"], cu["Y"])
    if cu["curve_type"] == "edwards":
        def add(P, Q):
            t = b * P[0] * Q[0] * P[1] * Q[1] % p
            return ((P[0] * Q[1] + P[1] * Q[0]) * pow(1 + t, -1, p) % p, (P[1] * Q[1] - a * P[0] * Q[0]) * pow(1 - t, -1, p) % p)
    else:
        def add(P, Q):
            if P == Q:
                m = (3 * P[0] * P[0] + a) * pow(2 * P[1], -1, p) % p
            else:
                m = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
            x = (m * m - P[0] - Q[0]) % p
            return (x, (m * (P[0] - x) - P[1]) % p)
    out, P = [], G
    for _ in range(count):
        out.append(P)
        P = add(P, G)
    return out


def fixture(curve, seed, records, custom=None):
    lib, pre, N, nb, radix, _, small_x = curveref.build(curve, custom)

    class Pt(ctypes.Structure):
        _fields_ = [("x", ctypes.c_uint64 * N), ("y", ctypes.c_uint64 * N), ("z", ctypes.c_uint64 * N)]
    PP = ctypes.POINTER(Pt)
    f = lambda name: getattr(lib, pre + name)
    for name, args in (("gen", [PP]), ("inf", [PP]), ("dbl", [PP]), ("neg", [PP]), ("cof", [PP]), ("add", [PP, PP]), ("sub", [PP, PP]), ("cpy", [PP, PP]),
                       ("mul", [ctypes.c_char_p, PP]), ("mul2", [ctypes.c_char_p, PP, ctypes.c_char_p, PP, PP])):
        f(name).argtypes = args
        f(name).restype = None
    f("isinf").argtypes = [PP]
    f("isinf").restype = ctypes.c_int
    H = lambda p: [[hex(v) for v in getattr(p, c)] for c in "xyz"]
    cp = lambda p: Pt.from_buffer_copy(bytes(p))
    ref = ctypes.byref
    rng = random.Random(seed)
    G = Pt()
    if not small_x:
        f("gen")(ref(G))
    else:
        # ecnXXXgen would take a square root (addchain); the same point from its affine coordinates (tests/golden/edwards_*.json /
        # weierstrass_*.json "gen", the reference's sign choice) through the reference's own nres and modone
        kind = "edwards" if curve.startswith(("ED", "NUMS256E")) else "weierstrass"
        gx, gy = gio.load("%s_%s.json" % (kind, curve))["gen"]
        U = ctypes.c_uint64 * N
        def limbs(v):
            return U(*[(v >> (radix * i)) & ((1 << radix) - 1) for i in range(N)])
        lib.nres.argtypes = [U, U]; lib.nres.restype = None
        lib.modone.argtypes = [U]; lib.modone.restype = None
        lib.nres(limbs(int(gx, 16)), G.x); lib.nres(limbs(int(gy, 16)), G.y); lib.modone(G.z)
    fx = {"curve": curve, "N": N, "Nbytes": nb, "radix": radix, "seed": seed, "gen": H(G), "source": "reference edwards.c / weierstrass.c + curve.py + generator-emitted field code, built by tests/golden/curveref.py"}
    recs = []
    P = cp(G)
    for k in range(records):
        e = bytes(rng.randrange(256) for _ in range(nb))
        g = bytes(rng.randrange(256) for _ in range(nb))
        if k == 1:
            e = (1).to_bytes(nb, "big")
        if k == 2:
            e = (0).to_bytes(nb, "big")
        if k == 3:
            e = b"\xff" * nb
        r = {"e": e.hex(), "f": g.hex(), "P": H(P)}
        M = cp(P); f("mul")(e, ref(M)); r["M"] = H(M)
        D = cp(M); f("dbl")(ref(D)); r["D"] = H(D)
        A = cp(M); f("add")(ref(D), ref(A)); r["A"] = H(A)
        S = cp(A); f("sub")(ref(D), ref(S)); r["S"] = H(S)
        Ng = cp(A); f("neg")(ref(Ng)); r["N"] = H(Ng)
        C = cp(A); f("cof")(ref(C)); r["C"] = H(C)
        R = Pt(); m2, d2 = cp(M), cp(D); f("mul2")(e, ref(m2), g, ref(d2), ref(R)); r["R"] = H(R)
        Z = cp(A); f("add")(ref(Ng), ref(Z)); r["A+N"] = H(Z); r["A+N_isinf"] = f("isinf")(ref(Z))      # P + (-P)
        T = cp(A); T2 = cp(A); f("add")(ref(T2), ref(T)); r["A+A"] = H(T)                                      # doubling through add
        r["isinf"] = [f("isinf")(ref(x)) for x in (M, D, A, R)]
        recs.append(r)
        P = cp(A) if k not in (2,) else cp(R)          # chain on; after the multiplication by zero continue from mul2's result
        if f("isinf")(ref(P)):
            P = cp(G)
    fx["records"] = recs
    O = Pt(); f("inf")(ref(O))
    sp = {"inf": H(O)}
    X = cp(O); f("dbl")(ref(X)); sp["dbl_inf"] = H(X)
    X = cp(G); f("add")(ref(O), ref(X)); sp["gen+inf"] = H(X)
    X = cp(O); f("add")(ref(G), ref(X)); sp["inf+gen"] = H(X)
    X = cp(O); f("mul")(bytes(rng.randrange(256) for _ in range(nb)), ref(X)); sp["mul_inf"] = H(X)
    fx["special"] = sp
    # ecnXXXset with BOTH coordinates (edwards.c:243-270, weierstrass.c:366-388): modimp, nres, the curve equation, modcmp -- no
    # square root, so the reference's own function runs.  Inputs: the affine points of the big-integer fixtures (edwards_*.json /
    # weierstrass_*.json "set_xy", on and off the curve)
    if custom is None:
        kind = "edwards" if curve.startswith(("ED", "NUMS256E")) else "weierstrass"
        aff = gio.load("%s_%s.json" % (kind, curve))
    else:
        pts = affine_multiples(custom, 6)
        hx = lambda v: v.to_bytes(nb, "big").hex()
        aff = {"set_xy": [{"x": hx(x), "y": hx(y), "valid": 1} for x, y in pts] + [{"x": hx(pts[1][0]), "y": hx((pts[1][1] + 1) % custom["p"]), "valid": 0}]}
        fx["custom"] = {k: (hex(v) if isinstance(v, int) and abs(v) > 1 << 32 else v) for k, v in custom.items()}
    f("set").argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, PP]
    f("set").restype = None
    sx = []
    for r in aff["set_xy"]:
        X = Pt(); f("set")(0, bytes.fromhex(r["x"]), bytes.fromhex(r["y"]), ref(X))
        sx.append({"x": r["x"], "y": r["y"], "P": H(X), "isinf": f("isinf")(ref(X))})
        assert f("isinf")(ref(X)) == (0 if r["valid"] else 1), "the big-integer model and the reference disagree on a point's validity"
    fx["set_xy"] = sx
    return fx


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for k, c in enumerate(CURVES):
        if only and c not in only:
            continue
        fx = fixture(c, 12000 + k, 10 if fx_small(c) else 6)
        gio.dump(fx, "curveref_%s.json" % c)
        print(c, len(fx["records"]), "records; gen x limb 0:", fx["gen"][0][0])
    # curves that are not in curve.py's table (modarith_amd.generate.EXAMPLE_CURVES), inserted the way curve.py asks its user to
    for k, (c, cu) in enumerate(custom_curves().items()):
        if only and c not in only:
            continue
        fx = fixture(c, 13000 + k, 8, custom=cu)
        gio.dump(fx, "curveref_%s.json" % c)
        print(c, len(fx["records"]), "records (custom curve); gen x limb 0:", fx["gen"][0][0])


def fx_small(c):
    return c in ("ED25519", "NIST256", "SECP256K1", "NUMS256E", "NUMS256W", "ED248")


if __name__ == "__main__":
    main()
