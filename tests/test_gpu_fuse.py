"""Fused chains on the GPU (modarith_amd/fuse.py): one kernel per chain, limbs equal to the call-by-call sequence of the
batched API bit for bit -- and through it to the reference's, since every batched call is pinned to the golden vectors --
on flat, tiled, unaligned, in-place and out-of-contract batches; values checked against plain integers after redc."""
import random

import numpy as np
import pytest

from tests.util import derive_any, random_soa, to_dev, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _accept(P):
    """the generators' acceptance chain (pseudo.py:1783-1796): z = 1 / ((x - y)(x + y))^2 on values already in internal form"""
    from modarith_amd.fuse import Chain
    ch = Chain(P, "accept")
    x, y = ch.inputs(2)
    s = ch.modsqr(ch.modmul(ch.modadd(x, y), ch.modsub(x, y)))
    ch.output(ch.modinv(s))
    return ch.build()


def _accept_calls(F, x, y):
    return F.modinv(F.modsqr(F.modmul(F.modadd(x, y), F.modsub(x, y))))


def _rand(fp, n, seed):
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << fp.radix, size=(fp.nlimbs, n), dtype=np.uint64)
    out[fp.nlimbs - 1] = rng.integers(0, 1 << (fp.n - fp.radix * (fp.nlimbs - 1)), size=n, dtype=np.uint64)
    return np.ascontiguousarray(out)


@pytest.mark.parametrize("P", ["X25519", "NIST256", "X448", "2519", "BP256"])
def test_acceptance_chain_equals_the_call_sequence(torch_cuda, P):
    torch = torch_cuda
    from modarith_amd.field import Field
    F, fp = Field(P), derive_any(P)
    f = _accept(P)
    n = 3 * 4096 + 1                                              # odd: the last element takes the 8-byte path where there is one
    x, y = to_dev(_rand(fp, n, 1)), to_dev(_rand(fp, n, 2))
    want = _accept_calls(F, x, y)
    z, = f(x, y)
    assert torch.equal(z, want)
    # values: 1 / ((x - y)(x + y))^2 on the plain integers behind the internal form
    m = 64
    xi, yi = F.to_ints(F.redc(x[:, :m].contiguous())), F.to_ints(F.redc(y[:, :m].contiguous()))
    got = F.to_ints(F.redc(z[:, :m].contiguous()))
    assert got == [pow(((a - b) * (a + b)) ** 2 % fp.p, -1, fp.p) if ((a - b) * (a + b)) % fp.p else 0 for a, b in zip(xi, yi)]
    # tiled batches, unaligned views (odd start: 8-byte path), in place
    xt, yt = F.to_tiled(x[:, :3 * 4096].contiguous(), 4096), F.to_tiled(y[:, :3 * 4096].contiguous(), 4096)
    zt, = f(xt, yt)
    assert torch.equal(F.to_flat(zt), want[:, :3 * 4096])
    zu, = f(x[:, 1:1000], y[:, 1:1000])
    assert torch.equal(zu, want[:, 1:1000])
    xc = x.clone()
    f(xc, y, out=[xc])
    assert torch.equal(xc, want)


def test_two_outputs_and_small_constants(torch_cuda):
    """a ladder-step-shaped chain (rfc7748.c:194-209: sums, differences, products, a multiplication by a small constant) with two
    results, one element pair per lane, two elements per lane on aligned batches"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    for P in ("X25519", "X448"):
        F, fp = Field(P), derive_any(P)
        ch = Chain(P, "step")
        a, b = ch.inputs(2)
        A, B = ch.modadd(a, b), ch.modsub(a, b)
        AA, BB = ch.modsqr(A), ch.modsqr(B)
        E = ch.modsub(AA, BB)
        ch.output(ch.modmul(AA, BB))
        ch.output(ch.modmul(E, ch.modadd(AA, ch.modmli(E, 121665))))
        f = ch.build()
        n = 1 << 16
        x, y = to_dev(_rand(fp, n, 5)), to_dev(_rand(fp, n, 6))
        o1, o2 = f(x, y)
        sA, sB = F.modadd(x, y), F.modsub(x, y)
        sAA, sBB = F.modsqr(sA), F.modsqr(sB)
        sE = F.modsub(sAA, sBB)
        assert torch.equal(o1, F.modmul(sAA, sBB))
        assert torch.equal(o2, F.modmul(sE, F.modadd(sAA, F.modmli(sE, 121665))))


def test_out_of_contract_limbs_take_the_exact_products(torch_cuda):
    """fabricated limbs (up to 2^64 - 1) in a few lanes: the vote fails for their waves, the chain runs the exact 128-bit products
    there and the split ones elsewhere -- the same words as the call-by-call sequence, whose first multiplication votes the same way"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    F, fp = Field("X25519"), derive_any("X25519")
    ch = Chain("X25519", "mulsqr")
    a, b = ch.inputs(2)
    ch.output(ch.modsqr(ch.modmul(a, b)))
    f = ch.build()
    n = 1 << 14
    x, y = _rand(fp, n, 7), _rand(fp, n, 8)
    rng = np.random.default_rng(9)
    for j in list(range(0, n, 997)) + list(range(4096, 4096 + 256)):
        x[:, j] = rng.integers(0, 1 << 64, size=fp.nlimbs, dtype=np.uint64)
    x, y = to_dev(x), to_dev(y)
    z, = f(x, y)
    assert torch.equal(z, F.modsqr(F.modmul(x, y)))


def test_a_ladder_step_with_selectors_and_lazy_forms(torch_cuda):
    """one Montgomery-ladder step as the reference writes it (rfc7748.c:190-223: modcsw on the scalar bit, generic=False sums,
    5 multiplications, 4 squarings, a24) as ONE kernel: five element batches and a per-element swap bit in, four batches out --
    equal to the same 22 calls of the batched API, limb for limb; the fused step moves 364 bytes per element, the calls 2 288"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    for P, a24 in (("X25519", 121665), ("X448", 39081)):
        F, fp = Field(P), derive_any(P)
        ch = Chain(P, "ladderstep")
        x1, x2, z2, x3, z3 = ch.inputs(5)
        sw = ch.selector()
        x2, x3 = ch.modcsw(sw, x2, x3)
        z2, z3 = ch.modcsw(sw, z2, z3)
        A, B, C, D = ch.modadd_lazy(x2, z2), ch.modsub_lazy(x2, z2), ch.modadd_lazy(x3, z3), ch.modsub_lazy(x3, z3)
        AA, BB, DA, CB = ch.modsqr(A), ch.modsqr(B), ch.modmul(D, A), ch.modmul(C, B)
        E = ch.modsub_lazy(AA, BB)
        for v in (ch.modmul(AA, BB), ch.modmul(E, ch.modadd_lazy(AA, ch.modmli(E, a24))),
                  ch.modsqr(ch.modadd_lazy(DA, CB)), ch.modmul(x1, ch.modsqr(ch.modsub_lazy(DA, CB)))):
            ch.output(v)
        with pytest.raises(ValueError, match="generic=False result"):
            ch.modadd_lazy(A, B)
        f = ch.build()
        n = 2 * 4096 + 3
        t = [to_dev(_rand(fp, n, 20 + i)) for i in range(5)]
        bit = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
        got = f(*t, bit)
        X1, X2, Z2, X3, Z3 = [x.clone() for x in t]
        F.modcsw(bit, X2, X3); F.modcsw(bit, Z2, Z3)
        sA, sB, sC, sD = F.modadd_lazy(X2, Z2), F.modsub_lazy(X2, Z2), F.modadd_lazy(X3, Z3), F.modsub_lazy(X3, Z3)
        sAA, sBB, sDA, sCB = F.modsqr(sA), F.modsqr(sB), F.modmul(sD, sA), F.modmul(sC, sB)
        sE = F.modsub_lazy(sAA, sBB)
        want = (F.modmul(sAA, sBB), F.modmul(sE, F.modadd_lazy(sAA, F.modmli(sE, a24))),
                F.modsqr(F.modadd_lazy(sDA, sCB)), F.modmul(X1, F.modsqr(F.modsub_lazy(sDA, sCB))))
        for k in range(4):
            assert torch.equal(got[k], want[k]), (P, k)
        assert ch.traffic_bytes() == 9 * 8 * fp.nlimbs + 4


def test_remaining_operations(torch_cuda):
    """modcmv, modnsqr, modhaf, modneg, modpro, modsqrt, nres / redc inside a chain against the same calls"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    P = "NIST256"
    F, fp = Field(P), derive_any(P)
    ch = Chain(P, "misc")
    a, b = ch.inputs(2)
    d = ch.selector()
    m = ch.modcmv(d, a, b)                     # d ? a : b
    q = ch.modnsqr(ch.nres(m), 3)
    h = ch.modneg(ch.modhaf(q))
    ch.output(ch.redc(h))
    ch.output(ch.modsqrt(ch.modsqr(a)))
    ch.output(ch.modpro(b))
    f = ch.build()
    n = 4096 + 5
    x, y = to_dev(_rand(fp, n, 31)), to_dev(_rand(fp, n, 32))
    sel = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
    o1, o2, o3 = f(x, y, sel)
    mm = y.clone(); F.modcmv(sel, x, mm)
    qq = F.nres(mm); F.modnsqr(qq, 3)
    hh = qq.clone(); F.modhaf(hh)
    assert torch.equal(o1, F.redc(F.modneg(hh)))
    assert torch.equal(o2, F.modsqrt(F.modsqr(x)))
    assert torch.equal(o3, F.modpro(y))


def test_fused_chain_and_generated_field_under_graph_capture(torch_cuda):
    """a fused chain and the kernels of a generated field are plain launches on the caller's stream: captured into a hipGraph on a
    side stream, replayed on fresh inputs (a launch-bound inner loop becomes one graph launch)"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    F, fp = Field("2519"), derive_any("2519")
    ch = Chain("2519", "graphed")
    u, v = ch.inputs(2)
    ch.output(ch.modmul(ch.modadd(u, v), ch.modsub(u, v)))
    f = ch.build()
    n = 4096
    x, y = to_dev(_rand(fp, n, 41)), to_dev(_rand(fp, n, 42))
    z, w = torch.empty_like(x), torch.empty_like(x)
    f(x, y, out=[z]); F.modsqr(z, out=w)                      # warm-up outside the capture (code objects, Field binding)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            f(x, y, out=[z])
            F.modsqr(z, out=w)
    torch.cuda.current_stream().wait_stream(side)
    x.copy_(to_dev(_rand(fp, n, 43))); y.copy_(to_dev(_rand(fp, n, 44)))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(w, F.modsqr(F.modmul(F.modadd(x, y), F.modsub(x, y))))


def test_c_consumer_of_a_fused_chain(torch_cuda, tmp_path):
    """examples/fused_chain.c: the chain's C entry point next to the four batched calls, from plain C"""
    import os
    import subprocess
    from modarith_amd.fuse import bench_chain
    bench_chain("X25519").build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fused_chain")
    plug, main = os.path.join(root, "modarith_amd", "plugins"), os.path.join(root, "modarith_amd")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "fused_chain.c"), "-I", os.path.join(root, "include"),
                           "-L", plug, "-l:libmodarith_amd_chain_bench_prod_X25519.so", "-L", main, "-l:libmodarith_amd.so",
                           "-Wl,-rpath," + plug, "-Wl,-rpath," + main, "-o", exe])
    p = subprocess.run([exe, "100001"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "limb for limb" in p.stdout, p.stdout[-500:] + p.stderr[-500:]


@pytest.mark.parametrize("P", ["X25519", "X448", "1305"])
def test_element_major_io(torch_cuda, P):
    """FusedChain.aos: the chain over element-major arrays x[n][Nlimbs] (how the scalar callers of field.c hold elements), transposed
    through LDS inside the kernel -- equal to aos_to_soa, the chain, soa_to_aos; whole chunks, a ragged tail, fewer elements than one
    chunk, a selector, an inversion"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    F, fp = Field(P), derive_any(P)
    ch = Chain(P, "aosio")
    a, b = ch.inputs(2)
    d = ch.selector()
    g, h = ch.modcsw(d, a, b)
    ch.output(ch.modmul(ch.modadd(g, h), ch.modsub(g, h)))
    ch.output(ch.modinv(h))
    f = ch.build()
    for n in (3 * 512, 2 * 512 + 77, 5, 1):
        x, y = to_dev(_rand(fp, n, 50 + n % 7)), to_dev(_rand(fp, n, 60 + n % 7))
        sel = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
        want = f(x, y, sel)
        xa, ya = F.to_aos(x), F.to_aos(y)
        got = f.aos(xa, ya, sel)
        for k in range(2):
            assert torch.equal(F.from_aos(got[k]), want[k]), (P, n, k)
        xin = xa.clone()
        f.aos(xa, ya, sel, out=[xa, ya])                      # in place on the element-major arrays
        assert torch.equal(xa, got[0]) and torch.equal(ya, got[1])
        del xin


def test_element_major_io_five_inputs(torch_cuda):
    """more input arrays than LDS images fit (five 5-limb arrays: one shared image, barriers between the arrays): the ladder step of
    test_a_ladder_step_with_selectors_and_lazy_forms over element-major arrays"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import Chain
    P, a24 = "X25519", 121665
    F, fp = Field(P), derive_any(P)
    ch = Chain(P, "ladderstep")
    x1, x2, z2, x3, z3 = ch.inputs(5)
    sw = ch.selector()
    x2, x3 = ch.modcsw(sw, x2, x3)
    z2, z3 = ch.modcsw(sw, z2, z3)
    A, B, C, D = ch.modadd_lazy(x2, z2), ch.modsub_lazy(x2, z2), ch.modadd_lazy(x3, z3), ch.modsub_lazy(x3, z3)
    AA, BB, DA, CB = ch.modsqr(A), ch.modsqr(B), ch.modmul(D, A), ch.modmul(C, B)
    E = ch.modsub_lazy(AA, BB)
    for v in (ch.modmul(AA, BB), ch.modmul(E, ch.modadd_lazy(AA, ch.modmli(E, a24))),
              ch.modsqr(ch.modadd_lazy(DA, CB)), ch.modmul(x1, ch.modsqr(ch.modsub_lazy(DA, CB)))):
        ch.output(v)
    f = ch.build()
    assert "sh[CH * SP]" in ch.source()                           # the shared image
    n = 3 * 512 + 201
    t = [to_dev(_rand(fp, n, 70 + i)) for i in range(5)]
    bit = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
    want = f(*t, bit)
    got = f.aos(*[F.to_aos(x) for x in t], bit)
    for k in range(4):
        assert torch.equal(F.from_aos(got[k]), want[k]), k


def test_empty_and_single_element_batches(torch_cuda):
    """n = 0 is a no-op (as every batched entry point), n = 1 takes the 8-byte path; both layouts"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from modarith_amd.fuse import bench_chain
    F, fp = Field("X25519"), derive_any("X25519")
    f = bench_chain("X25519").build()
    x, y = to_dev(_rand(fp, 1, 81)), to_dev(_rand(fp, 1, 82))
    z, = f(x, y)
    assert torch.equal(z, F.modsqr(F.modmul(F.modadd(x, y), F.modsub(x, y))))
    za, = f.aos(F.to_aos(x), F.to_aos(y))
    assert torch.equal(F.from_aos(za), z)
    e = torch.empty((5, 0), dtype=torch.int64, device="cuda")
    assert f(e, e)[0].shape == (5, 0)
    ea = torch.empty((0, 5), dtype=torch.int64, device="cuda")
    assert f.aos(ea, ea)[0].shape == (0, 5)
