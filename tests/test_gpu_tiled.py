"""The tiled limb-interleaved layout (include/modarith_amd.h "TILED": ld < n selects tiles of ld elements, each tile limb-
interleaved with stride ld): every per-prime field function must return, element for element, what it returns on the flat
layout.  The flat results are themselves checked against the reference's vectors and the oracle elsewhere; here the two
layouts are compared with each other, through the Python mirror and -- for a partial last tile and an odd n -- through the
C-ABI directly."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PRIMES = ["X25519", "NIST256", "X448", "SIDH751", "PM266M", "GM384"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("tile", [128, 4096])
@pytest.mark.parametrize("P", PRIMES)
def test_every_field_function_on_tiles(torch_cuda, P, tile):
    torch = torch_cuda
    from modarith_amd.field import Field
    F, T = Field(P, tile=None), Field(P, tile=tile)
    n = 3 * tile
    a, b = F.uniform(n, seed=7, array=1, plus_p=True), F.uniform(n, seed=7, array=2)
    A, B = T.to_tiled(a), T.to_tiled(b)
    assert A.shape == (3, F.N, tile) and torch.equal(T.to_flat(A), a)
    assert torch.equal(T.uniform(n, seed=7, array=2), B)                       # the generator writes tiles directly
    same = lambda x, y: torch.equal(T.to_flat(x), y)
    for op in ("modadd", "modsub", "modmul", "modadd_lazy", "modsub_lazy"):
        assert same(getattr(T, op)(A, B), getattr(F, op)(a, b)), op
    for op in ("modsqr", "modneg", "modneg_lazy", "nres", "redc", "modcpy", "modpro", "modinv", "modsqrt"):
        assert same(getattr(T, op)(A), getattr(F, op)(a)), op
    # in-place aliasing (out = input) on tiles
    C = A.clone()
    T.modmul(C, B, out=C)
    assert same(C, F.modmul(a, b))
    b0 = [int(v) for v in b[:, 5].cpu().numpy().view(np.uint64)]
    assert same(T.modmuls(A, b0), F.modmuls(a, b0))
    assert same(T.modmli(A, 121665), F.modmli(a, 121665))
    X, x = A.clone(), a.clone()
    T.modnsqr(X, 5); F.modnsqr(x, 5)
    assert same(X, x)
    h, Hh = F.modpro(a), T.modpro(A)
    assert same(T.modinv(A, Hh), F.modinv(a, h)) and same(T.modsqrt(A, Hh), F.modsqrt(a, h))
    assert torch.equal(T.modqr(None, A), F.modqr(None, a)) and torch.equal(T.modqr(Hh, A), F.modqr(h, a))
    for op in ("modfsb", "flatten"):
        X, x = A.clone(), a.clone()
        r1, r2 = getattr(T, op)(X), getattr(F, op)(x)
        f1, f2 = (r1[1], r2[1]) if isinstance(r1, tuple) else (r1, r2)
        assert same(X, x) and torch.equal(f1, f2), op
    X, x = A.clone(), a.clone()
    T.modhaf(X); F.modhaf(x)
    assert same(X, x)
    X, x = A.clone(), a.clone()
    T.modshl(3, X); F.modshl(3, x)
    assert same(X, x)
    r1, r2 = T.modshr(2, X), F.modshr(2, x)
    assert same(X, x) and torch.equal(r1 if not isinstance(r1, tuple) else r1[-1], r2 if not isinstance(r2, tuple) else r2[-1])
    for op in ("modis1", "modis0", "modsign", "modlimbs"):
        assert torch.equal(getattr(T, op)(A), getattr(F, op)(a)), op
    assert torch.equal(T.modcmp(A, B), F.modcmp(a, b)) and bool(T.modcmp(A, T.modadd(A, T.modzer(n))).all())
    assert same(T.modone(n), F.modone(n)) and same(T.modint(77, n), F.modint(77, n)) and same(T.mod2r(9, n), F.mod2r(9, n))
    d = (torch.arange(n, device="cuda") % 3 == 0).to(torch.int32)
    G1, F1, g1, f1 = A.clone(), B.clone(), a.clone(), b.clone()
    T.modcsw(d, G1, F1); F.modcsw(d, g1, f1)
    assert same(G1, g1) and same(F1, f1)
    T.modcmv(d, A, F1); F.modcmv(d, a, f1)
    assert same(F1, f1)
    # bytes <-> tiles
    by = F.modexp(a)
    assert torch.equal(T.modexp(A), by)
    I1, fl1 = T.modimp(by)
    I2, fl2 = F.modimp(by)
    assert same(I1, I2) and torch.equal(fl1, fl2) and I1.dim() == 3
    # element-major <-> tiles
    aos = F.to_aos(a)
    assert torch.equal(T.to_aos(A), aos) and same(T.from_aos(aos), a)
    # the time.c chains per lane
    z1, z2 = T.time_protocol("modmul", A, B, 1), F.time_protocol("modmul", a, b, 1)
    assert same(z1, z2)


@pytest.mark.parametrize("P", ["X25519", "X448"])
def test_partial_last_tile_and_odd_n_through_the_c_abi(torch_cuda, P):
    """n = 2 tiles + 77 elements (odd: the two-elements-per-lane path plus its one-lane tail) in buffers of three tiles"""
    torch = torch_cuda
    from modarith_amd import _lib
    from modarith_amd.field import Field
    lib, F = _lib.load(), Field(P, tile=None)
    tile, n = 256, 2 * 256 + 77
    a, b = F.uniform(3 * tile, seed=9, array=1), F.uniform(3 * tile, seed=9, array=2)
    A, B = F.to_tiled(a, tile), F.to_tiled(b, tile)
    C = torch.full_like(A, -1)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert getattr(lib, "modmul_%s_batch" % P)(A.data_ptr(), B.data_ptr(), C.data_ptr(), n, tile, st) == 0
    want = F.modmul(a, b)
    got = F.to_flat(C)
    assert torch.equal(got[:, :n], want[:, :n]) and bool((got[:, n:] == -1).all())          # nothing written past element n-1
    b0 = (ctypes.c_uint64 * F.N)(*[int(v) for v in b[:, 3].cpu().numpy().view(np.uint64)])
    C.fill_(-1)
    assert getattr(lib, "modmuls_%s_batch" % P)(A.data_ptr(), ctypes.cast(b0, ctypes.c_void_p), C.data_ptr(), n, tile, st) == 0
    got = F.to_flat(C)
    assert torch.equal(got[:, :n], F.modmuls(a, list(b0))[:, :n]) and bool((got[:, n:] == -1).all())
    # an unusable stride is refused
    assert getattr(lib, "modmul_%s_batch" % P)(A.data_ptr(), B.data_ptr(), C.data_ptr(), n, 100, st) != 0
    assert b"tiled layout" in lib.modarith_amd_last_error()
    assert getattr(lib, "modmul_%s_batch" % P)(A.data_ptr(), B.data_ptr(), C.data_ptr(), n, 64, st) != 0


def test_tiled_inputs_match_bulk_reference_digests(torch_cuda):
    """the reference's own outputs (tests/golden/bulk_digests.json) reached through the tiled layout"""
    torch = torch_cuda
    from modarith_amd.field import Field
    from tests.conftest import load_golden
    from tests.util import block_digests, bulk_inputs, to_dev, to_np
    g = load_golden("bulk_digests.json")
    for P in ("X25519", "X448"):
        T = Field(P, tile=4096)
        a, b = bulk_inputs(P, "edge", g["n"])
        A, B = T.to_tiled(to_dev(a)), T.to_tiled(to_dev(b))
        for op, r in (("modmul", T.modmul(A, B)), ("modsqr", T.modsqr(A)), ("redc", T.redc(A))):
            assert block_digests(to_np(T.to_flat(r)), g["block"]) == g["primes"][P]["edge"][op], (P, op)


def test_tiled_grid_stride_passes_under_a_small_grid_cap(torch_cuda):
    """with the default cap (65 536 workgroups) a tiled batch below 2^25 elements runs one chunk per workgroup; the
    grid-stride loop over tiles is exercised here in a child process with MA_MAX_BLOCKS_TILED=7 (the cap is process-static):
    every streaming kernel family on 5 tiles of 4096 + an odd partial tile, against the flat layout"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes, sys, torch
sys.path.insert(0, %r)
from modarith_amd import _lib
from modarith_amd.field import Field
lib = _lib.load()
for P in ("X25519", "X448"):
    F = Field(P, tile=None)
    tile, n = 4096, 5 * 4096 + 1237
    a, b = F.uniform(6 * tile, seed=3, array=1), F.uniform(6 * tile, seed=3, array=2)
    A, B = F.to_tiled(a, tile), F.to_tiled(b, tile)
    C = torch.full_like(A, -1)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda name: getattr(lib, "%%s_%%s_batch" %% (name, P))
    assert f("modmul")(A.data_ptr(), B.data_ptr(), C.data_ptr(), n, tile, st) == 0
    assert torch.equal(F.to_flat(C)[:, :n], F.modmul(a, b)[:, :n]), "modmul"
    assert f("modsqr")(A.data_ptr(), C.data_ptr(), n, tile, st) == 0
    assert torch.equal(F.to_flat(C)[:, :n], F.modsqr(a)[:, :n]), "modsqr"
    assert f("modmli")(A.data_ptr(), 121665, C.data_ptr(), n, tile, st) == 0
    assert torch.equal(F.to_flat(C)[:, :n], F.modmli(a, 121665)[:, :n]), "modmli"
    assert f("modinv")(A.data_ptr(), None, C.data_ptr(), n, tile, st) == 0
    assert torch.equal(F.to_flat(C)[:, :n], F.modinv(a)[:, :n]), "modinv"
    out = torch.empty(6 * tile, dtype=torch.int32, device="cuda")
    assert f("modis0")(A.data_ptr(), out.data_ptr(), n, tile, st) == 0
    assert torch.equal(out[:n], F.modis0(a)[:n]), "modis0"
print("tiled grid-stride ok")
''' % root
    env = dict(os.environ, MA_MAX_BLOCKS_TILED="7", MA_MAX_BLOCKS="5")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "tiled grid-stride ok" in p.stdout, p.stderr[-2000:]


@pytest.mark.parametrize("P,kind", [("X25519", 0), ("X448", 2)])
def test_headline_configuration_full_size_on_tiles_vs_oracle(oracle, torch_cuda, P, kind):
    """the benchmarked configuration itself: 2^24 elements in tiles of 4096 (one 512-element chunk per workgroup), modmul,
    EVERY element against the CPU oracle on all host cores"""
    import os
    torch = torch_cuda
    from modarith_amd.field import Field
    from tests.util import vp
    T = Field(P, tile=4096)
    n = 1 << 24
    A, B = T.uniform(n, seed=42, array=0), T.uniform(n, seed=42, array=1)          # bench.py's operands (rank 0)
    assert A.dim() == 3 and A.shape == (n // 4096, T.N, 4096)
    if P != "X25519":
        A, B = T.nres(A), T.nres(B)
    C = T.modmul(A, B)
    ha = np.ascontiguousarray(T.to_flat(A).cpu().numpy().view(np.uint64))
    hb = np.ascontiguousarray(T.to_flat(B).cpu().numpy().view(np.uint64))
    want = np.empty_like(ha)
    assert oracle.lib.oracle_parallel(kind, vp(ha), vp(hb), vp(want), n, n, len(os.sched_getaffinity(0))) == 0
    assert np.array_equal(T.to_flat(C).cpu().numpy().view(np.uint64), want), "%s modmul on tiles differs from the oracle at full size" % P
