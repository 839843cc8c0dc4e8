"""Build the reference's curve layer (edwards.c / weierstrass.c) for one curve, in THIS container only, from the reference's own
files: curve.py's statements executed from its source in a scratch directory (the ones that shell out to the generators are
skipped; the radix those would have returned comes from tests/golden/refgen.py running the same generator), the field code the
generator emits (every function that does not need the external `addchain` tool) pasted at the @field@ marker by curve.py's own
replace calls acting on scratch COPIES of edwards.c / weierstrass.c / curve.h.  The result is compiled with gcc and driven with
ctypes.  modpro / modinv / modqr / modsqrt are not emitted (refgen.py explains why), so ecnXXXget / ecnXXXset / ecnXXXaffine are
not callable; everything that stays projective is: inf, add, sub, dbl, neg, cpy, mul, mul2, cof, isinf -- and gen, except on
the curves whose generator has a small x (ecnXXXgen then recovers y with a square root: NUMS256E, ED248, ED376, ED500,
NUMS256W).  The field functions are emitted non-static so that the caller can also reach the reference's nres / modone.

No stand-in tool, header or generated file is written.  Nothing here ships: outputs go to a scratch directory; only vectors
(tests/golden/curveref_*.json) are committed.  Test tooling: never imported by the product or on the GPU box.
"""
import ast, contextlib, ctypes, io, os, shutil, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refgen  # noqa: E402

REF = refgen.REF


def build(curve: str, custom: dict = None):
    """-> (CDLL, prefix 'ecn_<curve>_', Nlimbs, Nbytes, radix, scratch dir, small_x)
    custom: a curve that is not in curve.py's table, given the way curve.py asks its user to insert one ("More curves can be added
    here", curve.py:73-203) -- the variables that block assigns: p, q, cof, prime_type ('pseudo' | 'monty'), curve_type ('edwards' |
    'weierstrass'), A, B, X, Y -- plus field_arg, the prime as the field generator takes it on its command line."""
    scratch = tempfile.mkdtemp(prefix="curveref_")
    for f in ("edwards.c", "weierstrass.c", "curve.h", "testcurve.c"):
        shutil.copy(os.path.join(REF, f), scratch)
    path = os.path.join(REF, "curve.py")
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"__name__": "__curveref__", "__file__": path}
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = ["curve.py", "64", curve]
    os.chdir(scratch)
    log = io.StringIO()
    field_done = False
    try:
        for node in tree.body:
            seg = ast.get_source_segment(src, node) or ""
            if custom is not None and isinstance(node, ast.If) and "This curve not supported" in seg:
                assert ns["p"] == 0, "%s is in curve.py's table" % curve
                ns.update(p=custom["p"], q=custom["q"], cof=custom["cof"], A=custom["A"], B=custom["B"], X=custom["X"], Y=custom["Y"],
                          prime_type=ns["PSEUDO"] if custom["prime_type"] == "pseudo" else ns["MONTY"],
                          curve_type=ns["EDWARDS"] if custom["curve_type"] == "edwards" else ns["WEIERSTRASS"])
                continue
            if "subprocess.run" in seg and "radix=" in seg.replace(" ", ""):
                # `radix = subprocess.run("python3 pseudo.py 64 <curve>").returncode`: run that generator through refgen instead
                script = "pseudo.py" if ns["prime_type"] == ns["PSEUDO"] else "monty.py"
                g = refgen.load(script, 64, custom["field_arg"] if custom is not None else curve)
                ns["radix"] = g["base"]
                open(os.path.join(scratch, "field.c"), "w").write(refgen.emit_c(g, makestatic=False))
                field_done = True
                continue
            if "subprocess" in seg and not isinstance(node, (ast.Import, ast.ImportFrom)):
                continue                      # the group-order generator run (group.c): not needed by the curve layer
            with contextlib.redirect_stdout(log):
                exec(compile(ast.Module([node], []), path, "exec"), ns)
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    assert field_done
    cfile = "edwards.c" if ns["curve_type"] == ns["EDWARDS"] else "weierstrass.c"
    so = os.path.join(scratch, "curve.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w", "-I", scratch, "-o", so, os.path.join(scratch, cfile)])
    small_x = "#define CONSTANT_X" in open(os.path.join(scratch, "curve.c")).read()
    return ctypes.CDLL(so, mode=os.RTLD_LAZY), "ecn_%s_" % curve.lower(), ns["limbs"], ns["Nbytes"], ns["radix"], scratch, small_x


if __name__ == "__main__":
    lib, pre, N, nb, radix, d, small_x = build(sys.argv[1] if len(sys.argv) > 1 else "ED25519")
    print(pre, N, nb, radix, d)
    class Pt(ctypes.Structure):
        _fields_ = [("x", ctypes.c_uint64 * N), ("y", ctypes.c_uint64 * N), ("z", ctypes.c_uint64 * N)]
    p = Pt()
    getattr(lib, pre + "gen")(ctypes.byref(p))
    print([hex(v) for v in p.x])
    getattr(lib, pre + "dbl")(ctypes.byref(p))
    print([hex(v) for v in p.x], [hex(v) for v in p.z])
