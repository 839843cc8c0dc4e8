"""Generator mode on the GPU (modarith_amd/generate.py): a field generated at run time computes what a built-in one does.
The example moduli also run through every parametrised parity test of test_gpu_parity / test_gpu_round2 / test_gpu_round3
(reference-generated vectors, generic oracle); here: the reference's self-test chain through the scalar entry points of the
plug-ins, a modulus generated ON THIS BOX during the test, and 2^255-19 generated under its unnamed tag against the built-in
X25519 kernels, limb for limb."""
import ctypes
import random

import numpy as np
import pytest

from tests.util import derive_any, generated_tags, random_soa, to_dev, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("P", generated_tags())
def test_scalar_abi_reference_selftest_chain_on_plugins(torch_cuda, P):
    """pseudo.py:1783-1796 / monty.py:2383-2396 through <fn>_<TAG>_ct of the plug-in (host pointers, reference signatures)"""
    from modarith_amd import _lib
    lib = _lib.load_plugin(P)
    fp = derive_any(P)
    U = ctypes.c_uint64 * fp.nlimbs
    f = lambda name: getattr(lib, "%s_%s_ct" % (name, P))
    rng = random.Random(199)
    for _ in range(6):
        x, y = rng.randrange(0, 2 * fp.p), rng.randrange(0, 2 * fp.p)
        want = pow(((x - y) * (x + y)) ** 2 % fp.p, -1, fp.p)
        ax, ay, at, az = U(*fp.to_limbs(x)), U(*fp.to_limbs(y)), U(), U()
        f("nres")(ax, ax); f("nres")(ay, ay)
        f("modadd")(ax, ay, at); f("modsub")(ax, ay, az)
        f("modmul")(at, az, ax); f("modsqr")(ax, az)
        f("modinv")(az, None, az)
        f("modsqrt")(az, None, az); f("modsqr")(az, az)
        f("modhaf")(az); f("modadd")(az, az, az)
        f("modshl")(1, az); f("modshr")(1, az)
        f("redc")(az, az)
        assert fp.from_limbs(list(az)) == want


def test_unnamed_25519_equals_the_builtin_x25519_kernels(torch_cuda, tmp_path, monkeypatch):
    """`generate 64 2**255-19` -> tag 25519 (pseudo.py:1942): same constants as the built-in X25519, so every limb of every
    function output must be the same, on canonical, < 2p and tiled batches (generated on this box, in a scratch directory)"""
    from modarith_amd import generate as gen
    from modarith_amd.field import Field
    monkeypatch.setattr(gen, "PLUGIN_DIR", str(tmp_path))
    F = Field.generate("2**255-19")
    B = Field("X25519")
    assert F.prime == "25519" and (F.N, F.radix, F.nbytes) == (B.N, B.radix, B.nbytes)
    n = 3 * 4096
    a, b = to_dev(random_soa("X25519", n, 11)), to_dev(random_soa("X25519", n, 12))
    for op in ("modmul", "modadd", "modsub"):
        assert np.array_equal(to_np(getattr(F, op)(a, b)), to_np(getattr(B, op)(a, b))), op
    for op in ("modsqr", "modneg", "nres", "redc", "modpro"):
        assert np.array_equal(to_np(getattr(F, op)(a)), to_np(getattr(B, op)(a))), op
    assert np.array_equal(to_np(F.modinv(a)), to_np(B.modinv(a)))
    assert np.array_equal(to_np(F.modmli(a, 121665)), to_np(B.modmli(a, 121665)))
    ta, tb = F.to_tiled(a, 4096), F.to_tiled(b, 4096)
    assert np.array_equal(to_np(F.to_flat(F.modmul(ta, tb))), to_np(B.modmul(a, b)))


def test_a_modulus_generated_on_this_box(torch_cuda, tmp_path, monkeypatch):
    """2^127 - 1 (three 43-bit limbs; no fixture, never built before): derive, compile, load, compute -- checked against plain
    integer arithmetic after redc, with non-canonical (< 2p) inputs, and with the acceptance chain of the reference's self-test"""
    from modarith_amd import generate as gen
    from modarith_amd.field import Field
    monkeypatch.setattr(gen, "PLUGIN_DIR", str(tmp_path))
    p = 2**127 - 1
    F = Field.generate("2**127-1")
    assert F.prime == "1271" and F.params.p == p and gen.installed(str(tmp_path))[0]["tag"] == "1271"
    assert Field("2**127-1").prime == "1271"                          # an expression binds the generated field of that tag
    rng = random.Random(5)
    n = 4096 + 77
    xs = [rng.randrange(0, 2 * p) for _ in range(n)]
    ys = [rng.randrange(0, 2 * p) for _ in range(n)]
    xs[:4] = [0, 1, p - 1, 2 * p - 1]
    x, y = F.nres(F.from_ints(xs)), F.nres(F.from_ints(ys))
    assert F.to_ints(F.redc(F.modmul(x, y))) == [(a * b) % p for a, b in zip(xs, ys)]
    assert F.to_ints(F.redc(F.modsqr(x))) == [(a * a) % p for a in xs]
    assert F.to_ints(F.redc(F.modadd(x, y))) == [(a + b) % p for a, b in zip(xs, ys)]
    assert F.to_ints(F.redc(F.modsub(x, y))) == [(a - b) % p for a, b in zip(xs, ys)]
    assert F.to_ints(F.redc(F.modmli(x, 121665))) == [(a * 121665) % p for a in xs]
    assert F.to_ints(F.redc(F.modinv(x))) == [pow(a, p - 2, p) for a in xs]
    t = F.modsqr(F.modmul(F.modadd(x, y), F.modsub(x, y)))
    assert F.to_ints(F.redc(F.modinv(t))) == [pow(((a - b) * (a + b)) ** 2 % p, p - 2, p) for a, b in zip(xs, ys)]
    by = F.modexp(x)                                               # big-endian bytes of the canonical value
    assert [int.from_bytes(bytes(r), "big") for r in by.cpu().numpy()] == [a % p for a in xs]


def test_c_consumer_of_a_generated_field(torch_cuda, tmp_path):
    """examples/generated_field.c: MODARITH_AMD_DECLARE(2519) + the plug-in on the link line, from plain C"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "generated_field")
    plug, main = os.path.join(root, "modarith_amd", "plugins"), os.path.join(root, "modarith_amd")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "generated_field.c"), "-I", os.path.join(root, "include"),
                           "-L", plug, "-l:libmodarith_amd_2519.so", "-L", main, "-l:libmodarith_amd.so",
                           "-Wl,-rpath," + plug, "-Wl,-rpath," + main, "-o", exe])
    for n in ("1000", "70001"):                          # one inversion per element; inversions shared between elements
        p = subprocess.run([exe, n], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "equal to the scalar entry points" in p.stdout, p.stdout[-500:] + p.stderr[-500:]


def test_more_than_2_32_elements_in_one_batch(torch_cuda):
    """Maximum sizes: 288 GB of HBM hold batches whose ELEMENT index no longer fits 32 bits.  2^32 + 8192 elements of the
    three-limb field 2^130 - 5 (103 GB per array, two arrays) through moduniform, modmul and modsqr in one launch each, flat
    and tiled; then EVERY element again through launches of 2^26 elements on views of the same arrays (other base addresses, other
    grid, same limb stride) -- equal bit for bit -- and windows around 0, 2^31, 2^32 and the end regenerated from (seed, j) alone
    and recomputed as small batches of their own."""
    torch = torch_cuda
    from modarith_amd.field import Field
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    n = (1 << 32) + 8192
    need = 2 * 3 * 8 * n + (8 << 30)
    if free < need:
        pytest.skip("needs %.0f GB of free HBM (free: %.0f GB)" % (need / 1e9, free / 1e9))
    F = Field("1305", tile=None)
    a = F.uniform(n, seed=9, array=1)
    c = F.modmul(a, a)
    step = 1 << 26
    for s in range(0, n, step):
        e = min(n, s + step)
        assert torch.equal(F.modmul(a[:, s:e], a[:, s:e]), c[:, s:e]), "modmul chunk at %d" % s
    F.modsqr(a, out=c)                                                     # same values by another kernel, over the same > 2^32 indices
    for s in range(0, n, step):
        e = min(n, s + step)
        assert torch.equal(F.modmul(a[:, s:e], a[:, s:e]), c[:, s:e]), "modsqr chunk at %d" % s
    for first in (0, (1 << 31) - 64, (1 << 32) - 64, n - 128):
        w = F.uniform(128, seed=9, array=1, first=first)
        assert torch.equal(w, a[:, first:first + 128]), "moduniform window at %d" % first
        assert torch.equal(F.modsqr(w), c[:, first:first + 128]), "window at %d" % first
    # the tiled layout of the same batch: [n / 4096, 3, 4096] -- the address formula beyond 2^32 elements
    del c
    T = Field("1305", tile=4096)
    ta = T.uniform(n, seed=9, array=1)
    assert ta.dim() == 3
    for s in range(0, n, step):
        e = min(n, s + step)
        assert torch.equal(ta[s // 4096:e // 4096].permute(1, 0, 2).reshape(3, e - s), a[:, s:e]), "tiled moduniform chunk at %d" % s
    del a
    tc = T.modsqr(ta)
    for first in (0, (1 << 31) - 4096, (1 << 32) - 4096, n - 4096):
        w = F.uniform(4096, seed=9, array=1, first=first)
        assert torch.equal(F.modsqr(w), tc[first // 4096]), "tiled window at %d" % first
    del ta, tc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("P", generated_tags())
def test_time_protocol_check_words_of_generated_fields(torch_cuda, P):
    """the generators end by running time.c on the prime they were given: its dependent chains on the seed-42 operands
    (pseudo.py:1235-1250, 1862-1866) per lane on the GPU; check words and redc'd limbs equal the reference's at depth 10^3 / 10^5,
    and the command-line report (--time) prints the same words"""
    from modarith_amd import generate as gen
    from modarith_amd.field import Field
    from tests.conftest import limbs, load_golden
    F = Field(P)
    fp = F.params
    g = load_golden("field_%s.json" % P)["time"]
    mk = lambda v: [(int(v, 16) >> (fp.radix * i)) & ((1 << fp.radix) - 1) for i in range(fp.nlimbs)]
    lanes = 70
    x, y, xs = F.from_limbs([mk(g["ra"])] * lanes), F.from_limbs([mk(g["rb"])] * lanes), F.from_limbs([mk(g["rs"])] * lanes)
    for outer, tag in ((1, "1k"), (100, "100k")):
        z = F.to_limbs(F.time_protocol("modmul", x, y, outer))
        assert z == [limbs(g["modmul_z_" + tag])] * lanes
        z = F.to_limbs(F.time_protocol("modsqr", xs, None, outer))
        assert z == [limbs(g["modsqr_z_" + tag])] * lanes
    rep = gen.time_report(P, outer=100, lanes=4096)
    assert rep[0].startswith("modmul check 0x%06x " % int(g["modmul_check_100k"], 16)), rep
    assert rep[1].startswith("modsqr check 0x%06x " % int(g["modsqr_check_100k"], 16)), rep
    assert rep[2].startswith("modinv check 0x"), rep


@pytest.mark.parametrize("name", ["M383", "T2519"])
def test_generated_montgomery_ladder(torch_cuda, name):
    """rfc7748() for a Montgomery curve other than X25519 / X448 (the #ifdef block a user of rfc7748.c adds, rfc7748.c:118-132): M-383
    over the built-in PM383 field and a ladder over the GENERATED field 2^251-9 -- batched and through the scalar entry point --
    against RFC 7748 section 5 in plain integers: clamping, masking of u, non-canonical u, u = 0, the base point, a DH exchange"""
    import ctypes
    import json
    import os
    import random
    torch = torch_cuda
    from modarith_amd import generate as gen
    from modarith_amd.field import rfc7748
    c = next(x for x in gen.EXAMPLE_LADDERS if x["name"] == name)
    m = json.load(open(os.path.join(gen.PLUGIN_DIR, "ladder_%s.json" % name)))
    nb, bits, a24, cof = m["nbytes"], m["nbits"], c["a24"], c["cof"]
    p = derive_any(c["field"]).p

    def model(k, u):
        k = int.from_bytes(k, "little"); u = int.from_bytes(u, "little")
        k &= ~((1 << cof) - 1); k &= (1 << bits) - 1; k |= 1 << (bits - 1)
        u = (u & ((1 << bits) - 1)) % p
        x1, x2, z2, x3, z3, swap = u, 1, 0, u, 1, 0
        for t in range(bits - 1, -1, -1):
            kt = (k >> t) & 1
            swap ^= kt
            if swap: x2, x3, z2, z3 = x3, x2, z3, z2
            swap = kt
            A, B, C, D = (x2 + z2) % p, (x2 - z2) % p, (x3 + z3) % p, (x3 - z3) % p
            AA, BB, DA, CB = A * A % p, B * B % p, D * A % p, C * B % p
            E = (AA - BB) % p
            x3, z3 = (DA + CB) ** 2 % p, x1 * (DA - CB) ** 2 % p
            x2, z2 = AA * BB % p, E * (AA + a24 * E) % p
        if swap: x2, z2 = x3, z3
        return (x2 * pow(z2, p - 2, p) % p).to_bytes(nb, "little")

    rng = random.Random(5)
    n = 70
    ks = [bytes(rng.randrange(256) for _ in range(nb)) for _ in range(n)]
    us = [bytes(rng.randrange(256) for _ in range(nb)) for _ in range(n)]
    us[0] = (12).to_bytes(nb, "little")                        # M-383's base point
    us[1] = (0).to_bytes(nb, "little")
    us[2] = (p + 5).to_bytes(nb, "little") if p + 5 < 1 << (8 * nb) else us[2]      # non-canonical u
    us[3] = b"\xff" * nb                                        # bits above Nbits are masked
    dev = lambda rows: torch.tensor([list(r) for r in rows], dtype=torch.uint8, device="cuda")
    got = rfc7748(name, dev(ks), dev(us)).cpu().numpy()
    assert [bytes(r) for r in got] == [model(k, u) for k, u in zip(ks, us)]
    # Diffie-Hellman on the curve: both sides reach the same bytes; and the scalar entry point (the reference's signature)
    base = dev([us[0]] * 2)
    pk = rfc7748(name, dev(ks[:2]), base)
    s1 = rfc7748(name, dev([ks[0]]), pk[1:2].contiguous())
    s2 = rfc7748(name, dev([ks[1]]), pk[0:1].contiguous())
    assert torch.equal(s1, s2)
    lib = ctypes.CDLL(gen.ladder_plugin_path(name))
    out = ctypes.create_string_buffer(nb)
    getattr(lib, "rfc7748_%s" % name)(ks[5], us[5], out)
    assert out.raw == model(ks[5], us[5])
