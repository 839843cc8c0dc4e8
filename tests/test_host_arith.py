"""The device arithmetic of the fused kernels, compiled for the HOST and run against the CPU oracle (no GPU needed):
tools/fe_host_check.hip instantiates the very functions the kernels wrap -- x25519_fe26_one, x448_fe28_one (ladders),
ed25519_mul_get_one, ed25519_mul2_get_one, ed448_mul_get_one (fused scalar / double multiplication + affine export) and the half-limb column products of
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
               [(0, 13, 13, 0, 0), (0, 8, 8, 0, 0), (0, 6, 15, 0, 0), (1, 4, 0, 0, 0), (3, 13, 3, 3, 15), (3, 13, 3, 3, 6), (3, 10, 10, 6, 15), (3, 10, 2, 15, 2)]),
              (1, 2**256 - 2**32 - 977, False,
               [(0, 4, 4, 0, 0), (0, 8, 1, 0, 0), (1, 2, 0, 0, 0), (1, 4, 0, 0, 0), (3, 3, 2, 1, 3), (3, 1, 3, 2, 2), (3, 2, 3, 3, 3), (3, 2, 4, 1, 8)]))
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
