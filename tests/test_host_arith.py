"""The device arithmetic of the fused kernels, compiled for the HOST and run against the CPU oracle (no GPU needed):
tools/fe_host_check.hip instantiates the very functions the kernels wrap -- x25519_fe26_one, x448_fe28_one (ladders),
Ed26Lad / Ed28Lad::mul_get_one and mulgen2_get_one (the ladder forms of the fused multiplications, with every exceptional class of input), ed25519 / ed448_mul2_get_straus_one (the Straus forms of the double multiplication), the fixed-base forms and the half-limb column products of
Field<P_X25519,true>, Field<P_NIST256,true> and Field<P_X448,true> -- with MA_DEV = __host__ __device__, and compares every output with the oracle
(rfc7748, ecn mul + ecn get, modmul / modsqr), including special points, corner scalars and the limb contract's edge
classes.  This checks the limb arithmetic and the group-law logic; code generation for gfx950 is checked on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_device_arithmetic_on_host_against_oracle(oracle, tmp_path):
    exe = str(tmp_path / "fe_host_check")
    odir = os.path.join(ROOT, "oracle")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", os.path.join(ROOT, "tools", "fe_host_check.hip"), "-o", exe, "--offload-arch=gfx950",
                    "-L" + odir, "-l:liboracle.so", "-Wl,-rpath," + odir], check=True, timeout=900)
    p = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if "records" in l]
    assert len(lines) == 31 and all(" 0 differ" in l for l in lines), p.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_secp256k1_fused_on_host_against_oracle(oracle, tmp_path):
    """the per-lane functions of the fused secp256k1 kernels (csrc/wn26.h on csrc/fk26.h), compiled for the host
    (tools/wn26_host.hip), against the oracle's ecn mul / mul2 followed by ecn get: random projective points, the point at
    infinity, scalars 0, 1, 2, the group order q, q +- 1, all ones, single windows; for mul2 also Q = +-P and f = e"""
    import ctypes
    import random
    so = str(tmp_path / "libwn26_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "wn26_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    U64 = ctypes.c_uint64
    C = "secp256k1"
    Pt, nb = oracle.ed[C]
    q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    rng = random.Random(11)
    be = lambda v: v.to_bytes(32, "big")
    words = lambda v: (U64 * 4)(*[(v >> (64 * k)) & (2**64 - 1) for k in range(4)])

    def point(kind):
        p = Pt()
        if kind == "inf":
            oracle.ecn(C, "inf")(ctypes.byref(p))
            return p
        oracle.ecn(C, "gen")(ctypes.byref(p))
        if kind == "rand":
            oracle.ecn(C, "mul")(be(rng.getrandbits(256)), ctypes.byref(p))      # projective, Z != 1
        return p

    def affine(p):
        x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        oracle.ecn(C, "get")(ctypes.byref(p), x, y)
        return x.raw, y.raw

    def out_bytes(xw, yw):
        return b"".join(int(xw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)), b"".join(int(yw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0))

    corner = [0, 1, 2, 7, 8, 9, 15, 16, 0x88, q - 1, q, q + 1, 2**256 - 1, 2**255, 2**256 - q, 2**256 - 8]
    for it in range(120):
        kind = "inf" if it % 16 == 1 else ("gen" if it % 16 == 4 else "rand")
        e = corner[it] if it < len(corner) else rng.getrandbits(256)
        p = point(kind)
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.secp256k1_mul_get_host(words(e), p.x, p.y, p.z, xw, yw)
        oracle.ecn(C, "mul")(be(e), ctypes.byref(p))
        assert out_bytes(xw, yw) == affine(p), ("mul_get", it)
    for it in range(60):
        e, f = rng.getrandbits(256), rng.getrandbits(256)
        p, qq = point("rand"), point("rand")
        if it % 8 == 1:
            p = point("inf")
        if it % 8 == 2:
            qq = point("inf")
        if it % 8 in (3, 4, 5):
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            if it % 8 != 4:
                oracle.ecn(C, "neg")(ctypes.byref(qq))                            # Q = -P
            if it % 8 == 3:
                f = e                                                            # e P + e (-P) = infinity -> (0, 1)
        if it == 6:
            e = f = 0
        if it == 7:
            e = f = 2**256 - 1
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.secp256k1_mul2_get_host(words(e), p.x, p.y, p.z, words(f), qq.x, qq.y, qq.z, xw, yw)
        r = Pt()
        oracle.ecn(C, "mul2")(be(e), ctypes.byref(p), be(f), ctypes.byref(qq), ctypes.byref(r))
        want = affine(r)
        assert out_bytes(xw, yw) == want, ("mul2_get", it)
        if it % 8 == 3:
            assert want == (be(0), be(1))


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_secp256k1_endomorphism_split_and_fused_forms_on_host(oracle, tmp_path):
    """csrc/glv26.h on the host: (1) the split k = k1 + k2 lambda (mod n) against Python integers -- identity and |k1|, |k2| < 2^128
    on corner scalars (0, 1, n, n +- 1, lambda, n - lambda, 2^128 +- 1, all ones) and 10^5 random ones; (2) the fused forms built on
    it (k P and e G + f Q as 128 doublings on one table) against the oracle's ecn mul / ecn mul2 followed by ecn get: random projective
    points, the point at infinity, the generator, Q = +-G, scalars that cancel"""
    import ctypes
    import random
    so = str(tmp_path / "libwn26_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "wn26_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    U64 = ctypes.c_uint64
    n = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    assert pow(lam, 3, n) == 1 and lam != 1
    rng = random.Random(17)
    words = lambda v: (U64 * 4)(*[(v >> (64 * k)) & (2**64 - 1) for k in range(4)])
    val = lambda w: sum(int(x) << (64 * i) for i, x in enumerate(w))
    corner = [0, 1, 2, 7, 8, 9, n - 1, n, n + 1, 2**256 - 1, 2**255, lam, n - lam, lam + 1, lam - 1, 2**128, 2**128 - 1, 2**128 + 1, 2**129,
              (n + 1) // 2, n // 2, 2**256 - n, int("8" * 64, 16), int("7" * 64, 16)]
    for it in range(100000):
        e = corner[it] if it < len(corner) else rng.getrandbits(256)
        k1, k2 = (U64 * 3)(), (U64 * 3)()
        sg = lib.secp256k1_glv_split_host(words(e), k1, k2)
        a, b = val(k1), val(k2)
        assert a < 2**128 and b < 2**128, hex(e)
        assert ((-a if sg & 1 else a) + (-b if sg & 2 else b) * lam - e) % n == 0, hex(e)

    C = "secp256k1"
    Pt, nb = oracle.ed[C]
    be = lambda v: v.to_bytes(32, "big")

    def point(kind):
        p = Pt()
        if kind == "inf":
            oracle.ecn(C, "inf")(ctypes.byref(p))
            return p
        oracle.ecn(C, "gen")(ctypes.byref(p))
        if kind == "rand":
            oracle.ecn(C, "mul")(be(rng.getrandbits(256)), ctypes.byref(p))
        if kind == "neg":
            oracle.ecn(C, "neg")(ctypes.byref(p))
        return p

    def affine(p):
        x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        oracle.ecn(C, "get")(ctypes.byref(p), x, y)
        return x.raw, y.raw

    out_bytes = lambda xw, yw: (b"".join(int(xw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)), b"".join(int(yw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)))
    for it in range(160):
        kind = "inf" if it % 16 == 1 else ("gen" if it % 16 == 4 else "rand")
        e = corner[it] if it < len(corner) else rng.getrandbits(256)
        p = point(kind)
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.secp256k1_glv_mul_get_host(words(e), p.x, p.y, p.z, xw, yw)
        oracle.ecn(C, "mul")(be(e), ctypes.byref(p))
        assert out_bytes(xw, yw) == affine(p), ("glv mul_get", it, hex(e))
    for it in range(80):
        e, f = rng.getrandbits(256), rng.getrandbits(256)
        kind = ("rand", "inf", "gen", "neg")[it % 4] if it < 40 else "rand"
        if it % 8 == 2:
            f = e                                   # e G + e G
        if it % 8 == 3:
            f = e                                   # with Q = -G: infinity
        if it % 8 == 5:
            e, f = corner[it % len(corner)], corner[(it * 7) % len(corner)]
        qq = point(kind)
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.secp256k1_glv_mulgen2_get_host(words(e), words(f), qq.x, qq.y, qq.z, xw, yw)
        g, r = point("gen"), Pt()
        oracle.ecn(C, "mul2")(be(e), ctypes.byref(g), be(f), ctypes.byref(qq), ctypes.byref(r))
        assert out_bytes(xw, yw) == affine(r), ("glv mulgen2_get", it, kind)
    for it in range(80):                             # e P + f Q, both split: P, Q random / infinite / equal / opposite
        e, f = rng.getrandbits(256), rng.getrandbits(256)
        p, qq = point("rand"), point("rand")
        if it % 8 == 1:
            p = point("inf")
        if it % 8 == 2:
            qq = point("inf")
        if it % 8 in (3, 4, 5):
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            if it % 8 != 4:
                oracle.ecn(C, "neg")(ctypes.byref(qq))
            if it % 8 == 3:
                f = e                                # e P + e (-P) = infinity -> (0, 1)
        if it % 8 == 6:
            e, f = corner[it % len(corner)], corner[(it * 5) % len(corner)]
        xw, yw = (U64 * 4)(), (U64 * 4)()
        if it in (23, 31, 39):                       # Q = P and f = n - e: the point at infinity, met as R = -Q inside the loop
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            e = rng.getrandbits(255) % n
            f = n - e
        if it in (47, 55):                           # Q = P, e = f small: the accumulator equals the table point at the first non-zero window
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            e = f = (3, 8)[it == 55]
        lib.secp256k1_glv_mul2_get_host(words(e), p.x, p.y, p.z, words(f), qq.x, qq.y, qq.z, xw, yw)
        r = Pt()
        oracle.ecn(C, "mul2")(be(e), ctypes.byref(p), be(f), ctypes.byref(qq), ctypes.byref(r))
        want = affine(r)
        assert out_bytes(xw, yw) == want, ("glv mul2_get", it)
        if it % 8 == 3:
            assert want == (be(0), be(1))


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_nist256_jacobian_fused_forms_on_host_against_oracle(oracle, tmp_path):
    """csrc/wj26.h + wn_affine.h on the host -- the kernels' pipeline for one record: P-256 in Jacobian coordinates on affine window tables,
    exceptional cases decided by the scalar and the last addition complete (k P, e G + f Q), detected per addition (e P + f Q) --
    against the oracle's ecn mul / ecn mul2 followed by ecn get.  The scalars are chosen for the exceptional cases: n - 2m and n + 2m
    (the accumulator meets +-Q at the last digit: m = 1..8), 0, 1..17, multiples of 16 (zero last digit), n, n +- 1, 2n - 2^256 .. 2^256 - 1
    (reduced mod n first), leading zero windows (accumulator at infinity for many digits), single non-zero digits anywhere, all
    nibbles 8 / 7 / 9; the points: random projective, the generator (Z = 1), the point at infinity"""
    import ctypes
    import random
    so = str(tmp_path / "libwn26_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "wn26_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    U64 = ctypes.c_uint64
    n = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
    C = "nist256"
    Pt, nb = oracle.ed[C]
    rng = random.Random(23)
    be = lambda v: v.to_bytes(32, "big")
    words = lambda v: (U64 * 4)(*[(v >> (64 * k)) & (2**64 - 1) for k in range(4)])

    def point(kind):
        p = Pt()
        if kind == "inf":
            oracle.ecn(C, "inf")(ctypes.byref(p))
            return p
        oracle.ecn(C, "gen")(ctypes.byref(p))
        if kind == "rand":
            oracle.ecn(C, "mul")(be(rng.getrandbits(256)), ctypes.byref(p))
        if kind == "neg":
            oracle.ecn(C, "neg")(ctypes.byref(p))
        return p

    def affine(p):
        x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
        oracle.ecn(C, "get")(ctypes.byref(p), x, y)
        return x.raw, y.raw

    out_bytes = lambda xw, yw: (b"".join(int(xw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)), b"".join(int(yw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)))
    scalars = list(range(0, 18)) + [n - 2 * m for m in range(1, 9)] + [n + 2 * m for m in range(1, 9)] + [n - m for m in (1, 3, 5, 15, 16, 17, 32)]
    scalars += [n, n + 1, 2**256 - 1, 2**256 - 2, 2**255, 2**256 - n, 2 * n - 2**256 + 5, 16, 32, 0x100, 0x880, 0x1000000]
    scalars += [d << (4 * i) for i in (1, 2, 31, 62, 63) for d in (1, 7, 8, 9, 15)] + [int("8" * 64, 16), int("7" * 64, 16), int("9" * 64, 16) % 2**256]
    scalars += [rng.getrandbits(b) for b in (8, 16, 64, 128, 200, 252) for _ in range(3)] + [rng.getrandbits(256) for _ in range(40)]
    for it, e in enumerate(scalars):
        kind = "inf" if it % 16 == 9 else ("gen" if it % 4 == 1 else "rand")
        p = point(kind)
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.nist256_jac_mul_get_host(words(e), p.x, p.y, p.z, xw, yw)
        oracle.ecn(C, "mul")(be(e), ctypes.byref(p))
        assert out_bytes(xw, yw) == affine(p), ("jacobian mul_get", it, hex(e), kind)
    # e G on the fixed-base table with the Jacobian mixed addition: top digit 0 / 1 / 2, e = d 2^256 mod n, single digits in every window
    gscalars = scalars + [2**256 - n, 2 * (2**256 - n), (2**256 - n) + 2**255, n - 2**255, n - 2**255 + 1, 2**255 + 2**254, 3 * 2**254 - 1, 3 * 2**254, 3 * 2**254 + 1,
                          int("10" * 128, 2), int("01" * 128, 2)]
    gscalars += [d << (5 * i) for i in range(0, 52, 3) for d in (1, 15, 16, 17, 31) if (d << (5 * i)) < 2**256]
    for e in gscalars:
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.nist256_jac_mulgen_get_host(words(e), xw, yw)
        p = point("gen")
        oracle.ecn(C, "mul")(be(e), ctypes.byref(p))
        assert out_bytes(xw, yw) == affine(p), ("jacobian mulgen_get", hex(e))
    for it in range(120):                            # e G + f Q: Q random / infinite / +-G, scalars cancelling, f from the exceptional list
        e, f = rng.getrandbits(256), rng.getrandbits(256)
        kind = ("rand", "inf", "gen", "neg")[it % 4] if it < 48 else "rand"
        if it % 8 in (2, 3):
            f = e                                    # Q = G: 2e G;  Q = -G: infinity
        if it % 8 == 5:
            f = scalars[(it * 3) % 60]
        if it % 8 == 6:
            e, f = scalars[it % 60], scalars[(it * 7) % 60]
        qq = point(kind)
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.nist256_jac_mulgen2_get_host(words(e), words(f), qq.x, qq.y, qq.z, xw, yw)
        g, r = point("gen"), Pt()
        oracle.ecn(C, "mul2")(be(e), ctypes.byref(g), be(f), ctypes.byref(qq), ctypes.byref(r))
        assert out_bytes(xw, yw) == affine(r), ("jacobian mulgen2_get", it, kind, hex(e), hex(f))
    for it in range(100):                            # e P + f Q: Jacobian accumulator, two affine tables, R = +-(table point) detected and redone completely
        e, f = rng.getrandbits(256), rng.getrandbits(256)
        p, qq = point("rand"), point("rand")
        if it % 8 == 1:
            p = point("inf")
        if it % 8 == 2:
            qq = point("inf")
        if it % 8 in (3, 4, 5):
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            if it % 8 != 4:
                oracle.ecn(C, "neg")(ctypes.byref(qq))
            if it % 8 == 3:
                f = e                                # e P + e (-P): the accumulator is at infinity after EVERY window
        if it % 8 == 6:
            e, f = scalars[it % 60], scalars[(it * 7) % 60]
        if it == 7:
            p, qq = point("inf"), point("inf")
        if it == 15:
            e = f = 0
        if it in (23, 31, 39):                       # Q = P and f = n - e: the sum is the point at infinity, met as R = -Q at the last addition of a window
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            e = rng.getrandbits(255) % n
            f = n - e
        if it in (47, 55):                           # Q = P, e = f small: the accumulator equals the table point at the first non-zero window
            oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(qq))
            e = f = (3, 8)[it == 55]
        xw, yw = (U64 * 4)(), (U64 * 4)()
        lib.nist256_jac_mul2_get_host(words(e), p.x, p.y, p.z, words(f), qq.x, qq.y, qq.z, xw, yw)
        r = Pt()
        oracle.ecn(C, "mul2")(be(e), ctypes.byref(p), be(f), ctypes.byref(qq), ctypes.byref(r))
        want = affine(r)
        assert out_bytes(xw, yw) == want, ("jacobian mul2_get on affine tables", it)
        if it % 8 == 3:
            assert want == (be(0), be(1))


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_nist256_affine_table_ignores_a_neighbour_off_the_curve(oracle, tmp_path):
    """csrc/wn_affine.h (round-5 advisor): two records in one lane's column share an inversion.  With (x, 0, Z != 0) -- off the curve,
    entries 2P, 4P, 6P, 8P have Z = 0 while entry P has not -- as the SECOND record, the first record's affine table must be what it is
    next to an ordinary neighbour (before the fix: 64 of its 64 x / y words came out different, all zero); flag(bad) stays 0."""
    import ctypes
    import random
    so = str(tmp_path / "libwn26_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "wn26_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    C = "nist256"
    Pt, nb = oracle.ed[C]
    rng = random.Random(84)

    def rand_point():
        p = Pt()
        oracle.ecn(C, "gen")(ctypes.byref(p))
        oracle.ecn(C, "mul")(rng.getrandbits(256).to_bytes(32, "big"), ctypes.byref(p))
        return p

    def table(p, q):
        t, fl = (ctypes.c_uint64 * 80)(), (ctypes.c_uint32 * 2)()
        lib.nist256_affine_table_pair_host(p.x, p.y, p.z, q.x, q.y, q.z, t, fl)
        return list(t), list(fl)

    for _ in range(4):
        good, other, bad = rand_point(), rand_point(), rand_point()
        for k in range(5):
            bad.y[k] = 0
        want, fl = table(good, other)
        assert fl == [0, 0] and any(want)
        got, fl = table(good, bad)
        assert fl == [0, 0]
        assert got == want
        inf = Pt()
        oracle.ecn(C, "inf")(ctypes.byref(inf))
        got, fl = table(good, inf)                            # (the documented case: a neighbour at infinity, flag bit 0)
        assert fl == [0, 1] and got == want


def test_lazy_limb_bounds_of_the_fused_weierstrass_fields(tmp_path):
    """fm26.h / fk26.h at the limb magnitudes wn26.h lets them reach (|limb| <= K 2^26 with the K of the comments there):
    products, squarings and two-product reductions of worst-case operands (all limbs at +-(K 2^26 - 1), alternating signs,
    random) against Python integers -- the random points of the other tests never come near these bounds"""
    import ctypes
    import random
    so = str(tmp_path / "libwn26_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "wn26_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    I32, U64 = ctypes.c_int32 * 10, ctypes.c_uint64 * 4
    rng = random.Random(7)
    val = lambda l: sum(int(x) << (26 * i) for i, x in enumerate(l))

    def operand(K, kind):
        top = K * (1 << 26) - 1
        if kind == 0:
            return [top] * 10
        if kind == 1:
            return [-top] * 10
        if kind == 2:
            return [top if i % 2 == 0 else -top for i in range(10)]
        if kind == 3:
            return [rng.choice((top, -top, 0, 1, -1)) for _ in range(10)]
        return [rng.randint(-top, top) for _ in range(10)]

    fields = ((0, 2**256 - 2**224 + 2**192 + 2**96 - 1, True,
               [(0, 13, 13, 0, 0), (0, 8, 8, 0, 0), (0, 6, 15, 0, 0), (1, 4, 0, 0, 0), (3, 13, 3, 3, 15), (3, 13, 3, 3, 6), (3, 10, 10, 6, 15), (3, 10, 2, 15, 2),
                (1, 13, 0, 0, 0), (1, 12, 0, 0, 0), (0, 10, 10, 0, 0), (0, 9, 4, 0, 0), (3, 3, 13, 8, 1), (3, 4, 5, 2, 1)]),                  # last row: the Jacobian forms of csrc/wj26.h
              (1, 2**256 - 2**32 - 977, False,
               [(0, 4, 4, 0, 0), (0, 8, 1, 0, 0), (1, 2, 0, 0, 0), (1, 4, 0, 0, 0), (3, 3, 2, 1, 3), (3, 1, 3, 2, 2), (3, 2, 3, 3, 3), (3, 2, 4, 1, 8),
                (1, 5, 0, 0, 0), (0, 5, 4, 0, 0), (0, 3, 7, 0, 0), (0, 2, 5, 0, 0), (3, 4, 5, 2, 1)]))        # last row: further pairs inside the bound (a Jacobian form that was tried)
    for which, p, mont, cases in fields:
        rinv2 = pow(pow(2, 286, p), -2, p) if mont else 1
        for mode, kf, kg, ku, kv in cases:
            for kind in range(5):
                for rep in range(1 if kind < 3 else 40):
                    f, g = operand(kf, kind), operand(kg or 1, (kind + rep) % 5)
                    u, v = operand(ku or 1, (kind + 1) % 5), operand(kv or 1, (kind + 2 + rep) % 5)
                    w = U64()
                    lib.wn26_field_product(which, mode, I32(*f), I32(*g), I32(*u), I32(*v), w)
                    got = sum(int(w[k]) << (64 * k) for k in range(4))
                    t = val(f) * val(g) if mode == 0 else (val(f) ** 2 if mode == 1 else val(f) * val(g) + val(u) * val(v))
                    assert got == t * rinv2 % p, (which, mode, kf, kg, ku, kv, kind, rep)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_fused_generator_multiplication_on_host_against_oracle(oracle, tmp_path):
    """wn26_mulgen_get_one (fixed-base tables generated/comb_<C>.h, complete mixed additions) for P-256 and secp256k1 on the host
    against the oracle's ecn gen + ecn mul + ecn get: corner scalars (0, 1, single windows, 8 / 9 in every nibble position, the
    group order and its neighbours, all ones) and random ones"""
    import ctypes
    import random
    so = str(tmp_path / "libwn26_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "wn26_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    U64 = ctypes.c_uint64 * 4
    rng = random.Random(13)
    orders = {"nist256": 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551,
              "secp256k1": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141}
    for which, C in enumerate(("nist256", "secp256k1")):
        Pt, nb = oracle.ed[C]
        q = orders[C]
        scalars = [0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88, 0x80, q - 1, q, q + 1, 2**256 - 1, 2**255, 2**256 - q, 2**256 - 8]
        scalars += [8 << (4 * i) for i in range(0, 64, 7)] + [9 << (4 * i) for i in range(3, 64, 9)] + [int("8" * 64, 16), int("7" * 64, 16), int("9" * 64, 16)]
        scalars += [rng.getrandbits(256) for _ in range(60)]
        for e in scalars:
            xw, yw = U64(), U64()
            lib.wn26_mulgen_get_host(which, U64(*[(e >> (64 * k)) & (2**64 - 1) for k in range(4)]), xw, yw)
            p = Pt()
            oracle.ecn(C, "gen")(ctypes.byref(p))
            oracle.ecn(C, "mul")(e.to_bytes(32, "big"), ctypes.byref(p))
            x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
            oracle.ecn(C, "get")(ctypes.byref(p), x, y)
            got = (b"".join(int(xw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)), b"".join(int(yw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)))
            assert got == (x.raw, y.raw), (C, hex(e))
        # G scalars per lane with one shared inversion (the kernels' form): groups that mix infinite and finite results
        want = {}

        def ref(e):
            if e not in want:
                p = Pt()
                oracle.ecn(C, "gen")(ctypes.byref(p))
                oracle.ecn(C, "mul")(e.to_bytes(32, "big"), ctypes.byref(p))
                x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
                oracle.ecn(C, "get")(ctypes.byref(p), x, y)
                want[e] = (x.raw, y.raw)
            return want[e]

        for G in (2, 4):
            groups = [[0] * G, [q] * G, [0, 5, q, 1][:G], [7, 0, 0, q][:G], [1, 2, 3, 0][:G], [q, q + 1, q - 1, 2 * 0][:G]]
            groups += [[scalars[(7 * i + 3 * g) % 40] for g in range(G)] for i in range(24)]
            for grp in groups:
                E_ = (ctypes.c_uint64 * (4 * G))(*[(e >> (64 * k)) & (2**64 - 1) for e in grp for k in range(4)])
                X_, Y_ = (ctypes.c_uint64 * (4 * G))(), (ctypes.c_uint64 * (4 * G))()
                lib.wn26_mulgen_get_many_host(which, G, E_, X_, Y_)
                for g, e in enumerate(grp):
                    got = (b"".join(int(X_[4 * g + k]).to_bytes(8, "big") for k in (3, 2, 1, 0)), b"".join(int(Y_[4 * g + k]).to_bytes(8, "big") for k in (3, 2, 1, 0)))
                    assert got == ref(e), (C, G, [hex(v) for v in grp], g)

        # e*G + f*Q (verification pattern): against the oracle's gen, mul2, get; Q random / infinite / +-G, scalars cancelling
        def point(kind):
            p = Pt()
            if kind == "inf":
                oracle.ecn(C, "inf")(ctypes.byref(p))
                return p
            oracle.ecn(C, "gen")(ctypes.byref(p))
            if kind == "rand":
                oracle.ecn(C, "mul")(rng.getrandbits(256).to_bytes(32, "big"), ctypes.byref(p))
            if kind == "neg":
                oracle.ecn(C, "neg")(ctypes.byref(p))
            return p

        for it in range(48):
            e, f = rng.getrandbits(256), rng.getrandbits(256)
            kind = ("rand", "inf", "gen", "neg", "rand", "rand")[it % 6]
            if it % 6 == 3:
                f = e                                   # e G + e (-G) = infinity
            if it % 6 == 4:
                e = 0
            if it % 6 == 5 and it > 20:
                f = 0
            Qp = point(kind)
            xw, yw = U64(), U64()
            lib.wn26_mulgen2_get_host(which, U64(*[(e >> (64 * k)) & (2**64 - 1) for k in range(4)]),
                                      U64(*[(f >> (64 * k)) & (2**64 - 1) for k in range(4)]), Qp.x, Qp.y, Qp.z, xw, yw)
            G, R = point("gen"), Pt()
            oracle.ecn(C, "mul2")(e.to_bytes(32, "big"), ctypes.byref(G), f.to_bytes(32, "big"), ctypes.byref(Qp), ctypes.byref(R))
            x, y = ctypes.create_string_buffer(nb), ctypes.create_string_buffer(nb)
            oracle.ecn(C, "get")(ctypes.byref(R), x, y)
            got = (b"".join(int(xw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)), b"".join(int(yw[k]).to_bytes(8, "big") for k in (3, 2, 1, 0)))
            assert got == (x.raw, y.raw), (C, "mulgen2", it)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_unsigned_limb_fields_at_the_edges_of_their_ranges(tmp_path):
    """fe26.h / fe28.h and the Edwards formulas of ed26.h / ed28.h (tools/fe_limb_host.hip, host build) with every limb drawn from
    {0, 1, max - 1, max, random} of the range the header documents for that operand -- tight values, sums of two tight values,
    the scaled operands of ed26.h -- against Python integers: value of the result mod p, and the range the header promises for
    it.  Unsigned limbs wrap silently when a subtrahend's limb exceeds minuend + k p; random field elements never come
    near that (probability 2^-55), a stashed canonical 0 does (round 4: Fe28::sub under the doubling of the order-4 point)."""
    import ctypes
    import random
    so = str(tmp_path / "libfe_limb_host.so")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", "-shared", "-fPIC", "--offload-host-only", os.path.join(ROOT, "tools", "fe_limb_host.hip"), "-o", so],
                   check=True, timeout=900)
    lib = ctypes.CDLL(so)
    rng = random.Random(2026)
    U32 = ctypes.c_uint32

    class Fld:
        def __init__(self, name, nl, p, pos, tight_hi):
            self.name, self.nl, self.p, self.pos, self.tight_hi = name, nl, p, pos, tight_hi
            self.op = getattr(lib, name + "_op")
            self.formula = getattr(lib, "ed" + name[2:] + "_formula")

        def value(self, limbs):
            return sum(int(v) << self.pos[i] for i, v in enumerate(limbs)) % self.p

        def scaled(self, scale2):            # exclusive upper bounds: scale2 x the tight range (ed26.h's scale s = scale2 / 2)
            return [h * scale2 for h in self.tight_hi]

        def draw(self, hi, style=None):
            style = rng.randrange(6) if style is None else style
            out = []
            for h in hi:
                k = style if style < 4 else rng.randrange(6)
                out.append([0, 1, h - 2, h - 1][k] if k < 4 else rng.randrange(h))
            return out

        def run(self, op, f, g=None):
            r = (U32 * self.nl)()
            self.op(op, (U32 * self.nl)(*f), (U32 * self.nl)(*(g if g is not None else f)), r)
            return list(r)

        def inside(self, limbs, hi):
            return all(v < h for v, h in zip(limbs, hi))

    p25519, p448 = 2**255 - 19, 2**448 - 2**224 - 1
    # tight as the headers define it: fe26 even < 2^26, odd < 2^25, limb 1 < 2^25 + 2^16;  fe28 < 2^28, limbs 1 and 9 < 2^28 + 2^9
    f26 = Fld("fe26", 10, p25519, [(51 * i + 1) // 2 for i in range(10)], [(1 << 26) if i % 2 == 0 else (1 << 25) + ((1 << 16) if i == 1 else 0) for i in range(10)])
    f28 = Fld("fe28", 16, p448, [28 * i for i in range(16)], [(1 << 28) + ((1 << 9) if i in (1, 9) else 0) for i in range(16)])
    ADD, SUB, MUL, SQR, MLI, MLA, WC, MULK, SQRK = range(9)

    def pairs(n):                              # every pure style against every pure style, then mixtures
        for a in range(4):
            for b in range(4):
                yield a, b
        for _ in range(n):
            yield None, None

    # ---- fe26 (scale 1 = twice tight, as ed26.h counts)
    T26 = f26.tight_hi
    S = lambda s: f26.scaled(int(2 * s))       # noqa: E731   scale s -> bounds
    for sa, sb in pairs(300):
        f, g = f26.draw(S(1.0), sa), f26.draw(T26, sb)                      # sub: minuend a sum of two tight values, subtrahend tight
        r = f26.run(SUB, f, g)
        assert f26.value(r) == (f26.value(f) - f26.value(g)) % p25519 and f26.inside(r, S(2.0)), ("fe26 sub", sa, sb)
        for fs, gs in ((2.0, 1.5), (1.5, 1.5), (2.5, 1.0), (1.0, 1.5), (2.0, 0.5)):     # the operand scales ed26.h multiplies
            f, g = f26.draw(S(fs), sa), f26.draw(S(gs), sb)
            r = f26.run(MUL, f, g)
            assert f26.value(r) == f26.value(f) * f26.value(g) % p25519 and f26.inside(r, T26), ("fe26 mul", fs, gs, sa, sb)
        f = f26.draw(S(1.5), sa)
        r = f26.run(SQR, f)
        assert f26.value(r) == f26.value(f) ** 2 % p25519 and f26.inside(r, T26), ("fe26 sqr", sa)
        f, g = f26.draw(S(1.5), sa), f26.draw(T26, sb)
        r = f26.run(MLA, f, g)
        assert f26.value(r) == (f26.value(f) * 121665 + f26.value(g)) % p25519 and f26.inside(r, [h + 64 for h in S(1.0)]), ("fe26 mul_small_add", sa, sb)
        r = f26.run(MLI, f)
        assert f26.value(r) == f26.value(f) * 121665 % p25519 and f26.inside(r, T26), ("fe26 mul_small", sa)
        f = f26.draw([1 << 31] * 10, sa)
        r = f26.run(WC, f)
        assert f26.value(r) == f26.value(f) and f26.inside(r, T26), ("fe26 wc", sa)

    # ---- fe28: sums of two tight values everywhere the Edwards formulas put them
    T28 = f28.tight_hi
    SUM28 = [2 * h - 1 for h in T28]
    for sa, sb in pairs(300):
        f, g = f28.draw(SUM28, sa), f28.draw(SUM28, sb)
        r = f28.run(SUB, f, g)
        assert f28.value(r) == (f28.value(f) - f28.value(g)) % p448 and f28.inside(r, T28), ("fe28 sub", sa, sb)
        r = f28.run(MUL, f, g)
        assert f28.value(r) == f28.value(f) * f28.value(g) % p448 and f28.inside(r, T28), ("fe28 mul", sa, sb)
        h15 = [h + (h - 1) // 2 for h in SUM28]                              # (xs + y) of add_cached: below 1.5 x 2^29
        f, g = f28.draw(SUM28, sa), f28.draw(h15, sb)
        r = f28.run(MUL, f, g)
        assert f28.value(r) == f28.value(f) * f28.value(g) % p448 and f28.inside(r, T28), ("fe28 mul 1.5", sa, sb)
        r = f28.run(SQR, f)
        assert f28.value(r) == f28.value(f) ** 2 % p448 and f28.inside(r, T28), ("fe28 sqr", sa)
        f, g = f28.draw(T28, sa), f28.draw(SUM28, sb)
        for a, b in ((f, g), (g, f)):
            r = f28.run(MULK, a, b)
            assert f28.value(r) == f28.value(f) * f28.value(g) % p448 and f28.inside(r, T28), ("fe28 mul_k", sa, sb)
        r = f28.run(SQRK, f)
        assert f28.value(r) == f28.value(f) ** 2 % p448 and f28.inside(r, T28), ("fe28 sqr_k", sa)
        g = f28.draw(T28, sb)
        r = f28.run(MLA, f, g)
        assert f28.value(r) == (f28.value(f) * 39081 + f28.value(g)) % p448 and f28.inside(r, [h + 64 for h in SUM28]), ("fe28 mul_small_add", sa, sb)
        r = f28.run(MLI, f)
        assert f28.value(r) == f28.value(f) * 39081 % p448 and f28.inside(r, T28), ("fe28 mul_small", sa)
        f = f28.draw([1 << 31] * 16, sa)
        r = f28.run(WC, f)
        assert f28.value(r) == f28.value(f) and f28.inside(r, T28), ("fe28 wc", sa)

    # ---- the formulas, on field elements (not curve points: the formulas are polynomial identities checked coordinate by coordinate)
    def coords(F, flat):
        return [F.value(flat[k * F.nl:(k + 1) * F.nl]) for k in range(4)]

    def formula(F, what, P, a, b=None, c=None):
        buf = (U32 * (4 * F.nl))(*[v for co in P for v in co])
        arr = lambda v, n: (U32 * n)(*v) if v is not None else (U32 * n)()        # noqa: E731
        F.formula(what, buf, arr(a, 4 * F.nl if len(a) == 4 * F.nl else F.nl) if a is not None else arr(None, F.nl), arr(b, F.nl), arr(c, F.nl))
        out = list(buf)
        assert all(F.inside(out[k * F.nl:(k + 1) * F.nl], F.tight_hi) for k in range(4)), (F.name, "formula output not tight", what)
        return coords(F, out)

    d25519 = 0x52036cee2b6ffe738cc740797779e89800700a4d4141d8ab75eb4dca135978a3
    for sa, sb in pairs(120):
        for F, p in ((f26, p25519), (f28, p448)):
            P = [F.draw(F.tight_hi, sa if k != 1 else sb) for k in range(4)]
            X, Y, Z, T = [F.value(v) for v in P]
            # doubling
            A, B, C = X * X, Y * Y, 2 * Z * Z
            if F is f26:                       # a = -1
                H = A + B; E = H - (X + Y) ** 2; G = A - B; Fv = C + G              # noqa: E702
            else:                              # a = 1
                G = A + B; E = (X + Y) ** 2 - G; Fv = G - C; H = A - B             # noqa: E702
            want = [E * Fv % p, G * H % p, Fv * G % p, E * H % p]
            assert formula(F, 0, P, None) == want, (F.name, "dbl", sa, sb)
            # addition of a cached affine operand
            if F is f26:
                yp, ym, t2d = F.draw(F.tight_hi, sb), F.draw(F.tight_hi, sa), F.draw(F.scaled(3), sb)      # t2d up to scale 1.5 (3 x tight)
                a, b, c, d = (Y - X) * F.value(ym), (Y + X) * F.value(yp), T * F.value(t2d), 2 * Z
                e, f_, g, h = b - a, d - c, d + c, b + a
                want = [e * f_ % p, g * h % p, f_ * g % p, e * h % p]
                assert formula(F, 1, P, yp, ym, t2d) == want, (F.name, "add_cached", sa, sb)
            else:
                xs, y, tds = F.draw(SUM28, sb), F.draw(F.tight_hi, sa), F.draw(SUM28, sb)
                A, B, Cc, D, M = X * F.value(xs), Y * F.value(y), T * F.value(tds), Z, (X + Y) * (F.value(xs) + F.value(y))
                E, Fv, G, H = M - A - B, D + Cc, D - Cc, B - A
                want = [E * Fv % p, G * H % p, Fv * G % p, E * H % p]
                assert formula(F, 1, P, xs, y, tds) == want, (F.name, "add_cached", sa, sb)
            # the Straus forms' addition of a projective cached table entry (ed26s.h / ed28s.h), both signs; entries are tight
            # (they come out of from_words)
            Q = [F.draw(F.tight_hi, sb if k != 2 else sa) for k in range(4)]
            q0, q1, q2, q3 = [F.value(v) for v in Q]
            flatQ = [v for co in Q for v in co]
            for what, sg in ((2, 1), (3, -1)):
                if F is f26:                   # (Y+X, Y-X, 2dT, 2Z); the negated entry swaps the sums and negates 2dT
                    yp_, ym_, t2_ = (q0, q1, q2) if sg == 1 else (q1, q0, -q2)
                    a, b, c, d = (Y - X) * ym_, (Y + X) * yp_, T * t2_, Z * q3
                    e, f_, g, h = b - a, d - c, d + c, b + a
                    want = [e * f_ % p, g * h % p, f_ * g % p, e * h % p]
                else:                          # (X, Y, 39081 T, Z); the negated entry negates X and 39081 T
                    A, B, Cc, D, M = X * sg * q0, Y * q1, T * sg * q2, Z * q3, (X + Y) * (sg * q0 + q1)
                    E, Fv, G, H = M - A - B, D + Cc, D - Cc, B - A
                    want = [E * Fv % p, G * H % p, Fv * G % p, E * H % p]
                assert formula(F, what, P, flatQ) == want, (F.name, "Straus add_pc", what, sa, sb)
