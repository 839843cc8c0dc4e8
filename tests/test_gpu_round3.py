"""GPU parity, round 3: the reference's corner-case protocol (edge.py / edge.c) through the HIP library for every built prime."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CORE = ["X25519", "NIST256", "X448"]
EXTRA = list(__import__("modarith_amd.emit", fromlist=["EXTRA_PRIMES"]).EXTRA_PRIMES) + __import__("tests.util", fromlist=["generated_tags"]).generated_tags()


class _GpuEdgeEngine:
    """all 17 pairs as one batch of 17 elements, through the batched C-ABI"""
    def __init__(self, F, nbytes):
        import torch
        self.F, self.nb, self.torch = F, nbytes, torch
    def imp(self, ints):
        raw = np.frombuffer(b"".join(int(v).to_bytes(self.nb, "big") for v in ints), dtype=np.uint8).reshape(len(ints), self.nb)
        a, _ = self.F.modimp(self.torch.from_numpy(raw.copy()).cuda())
        return a
    def inv(self, x): return self.F.modinv(x)
    def sqrt(self, x): return self.F.modsqrt(x)
    def add(self, x, y): return self.F.modadd(x, y)
    def sub(self, x, y): return self.F.modsub(x, y)
    def mul(self, x, y): return self.F.modmul(x, y)
    def sqr(self, x): return self.F.modsqr(x)
    def cmp(self, x, y): return self.F.modcmp(x, y).cpu().tolist()


@pytest.mark.parametrize("P", CORE + EXTRA)
def test_edge_protocol_gpu(P):
    """edge.py:341-358's corner pairs and, after modadd(x,x,x) / modadd(y,y,y) (edge.c:167), their doubled forms:
    1/(1/a), a+b, a-b, b-a, a*b, sqr(sqrt(sqr a)) -- each compared with modcmp against the imported expected value"""
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from modarith_amd.field import Field
    from tests.util import derive_any as derive
    from tests import edge_protocol
    fp = derive(P)
    assert edge_protocol.run(_GpuEdgeEngine(Field(P), fp.nbytes), fp.p, fp.n, fp.nbytes) == []


@pytest.mark.parametrize("name", ["NIST256", "SECP256K1", "ED25519"])
def test_mulgen_get_second_grid_stride_pass(name):
    """the fixed-base kernels give each lane four scalars and cap the grid at 262144 lanes: with n = 4 * 262144 + 77 the
    grid-stride loop runs a second (partial) pass, which the suite's other sizes never reach.  Checked against the general
    fused kernel on the generator, record by record."""
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from modarith_amd.edwards import Curve
    C = Curve(name)
    n = 4 * 262144 + 77
    g = torch.Generator(device="cuda").manual_seed(21)
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    gx, gy, gs = C.mulgen_get(e)
    # reference for the tail region and a strided sample of the first pass (the whole batch through mul_get would take the
    # general kernel through 1M scalar multiplications: fine on the GPU, a second or two)
    wx, wy, ws = C.mul_get(e, C.gen(n))
    assert torch.equal(gx, wx) and torch.equal(gy, wy) and torch.equal(gs, ws)


def _zero_forms(F):
    """limb vectors whose VALUE is zero: 0, p (top limb unmasked) and 2p (outside the functions' domain [0, 2p), where
    modis0 itself no longer sees a zero -- the shared product must still be protected from it)"""
    fp = F.params
    return [fp.to_limbs(0), fp.to_limbs(fp.p), fp.to_limbs(2 * fp.p)]


@pytest.mark.parametrize("P", CORE + ["SECP256K1", "NIST384", "PM266M", "M607", "C2065", "SIDH751"])
def test_simultaneous_inversion_equals_one_modinv_per_element(P):
    """modinv_<P>_batch on a large batch shares one inversion between up to 64 elements (kernels.h k_inv_simul; fields of at
    most 9 limbs and radix <= 60, the others keep one inversion per element).  Its words must be those of the per-element
    kernel (both return the normalised form nres(redc(1/x))) for every element of the domain -- also for elements whose
    value is zero in any representation (they must not poison their group; modinv(0) = 0), in place, and on tiles -- and a
    fabricated out-of-contract element (whose own result is as undefined as in the reference) must leave every other
    element of its group untouched."""
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from modarith_amd.field import Field
    F = Field(P)
    N = F.N
    n = 3 * 16384 + 1237
    x = F.nres(F.uniform(n, seed=11, array=3))
    zf = _zero_forms(F)
    idx = list(range(0, n, 997)) + [1, 2, 3, n - 1, 16384, 16385, 2 * 16384 + 5]
    for k, j in enumerate(idx):
        x[:, j] = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in zf[k % len(zf)]], dtype=torch.int64)

    def per_element(v):
        w = torch.empty_like(v)
        for lo in range(0, v.shape[1], 8192):
            hi = min(v.shape[1], lo + 8192)
            w[:, lo:hi] = F.modinv(v[:, lo:hi].contiguous())
        return w
    want = per_element(x)
    got = F.modinv(x)
    assert torch.equal(got, want), P
    # value check: x * (1/x) = 1 where x != 0 (mod p), 1/x = 0 where x = 0
    z0 = torch.zeros(n, dtype=torch.bool, device="cuda")
    z0[idx] = True
    assert bool((F.modis0(got).bool() == z0).all())
    one = F.redc(F.modmul(x, got))
    assert torch.equal(one[:, ~z0], F.redc(F.modone(n))[:, ~z0])
    # in place (prefix products go to scratch) and on tiles
    y = x.clone()
    F.modinv(y, out=y)
    assert torch.equal(y, want)
    m = (n // 4096) * 4096
    T = Field(P, tile=4096)
    assert torch.equal(T.to_flat(T.modinv(T.to_tiled(x[:, :m].contiguous()))), want[:, :m])
    # normalised form: the words depend on the value only -- the progenitor form returns them too
    h = F.modpro(x[:, :4096].contiguous())
    assert torch.equal(F.modinv(x[:, :4096].contiguous(), h), want[:, :4096])
    # fabricated limbs (outside the limb contract) in two elements: they get their own inversion on the exact products, as the
    # per-element kernel gives a wave that holds one, and change nothing else -- all words equal again
    bad = x.clone()
    bad[0, 5000] = -1
    bad[N - 1, 20000] = 1 << 62
    assert torch.equal(F.modinv(bad), per_element(bad))


@pytest.mark.parametrize("P", ["X25519", "X448", "NIST521", "SIDH751", "C2065"])
def test_lds_transposed_converters(P):
    """element-major <-> limb-interleaved through the LDS-staged kernels (16-byte-aligned buffers, even stride, <= 14 limbs):
    chunk boundaries (512 elements), a short and odd last chunk, a padded even stride, tiles of 128 and 4096 -- against
    torch's own transposition; odd strides and unaligned views take the plain kernels and must agree too"""
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from modarith_amd.field import Field, tile_batch
    F = Field(P)
    N = F.N
    g = torch.Generator(device="cuda").manual_seed(17)
    for n in (2, 510, 512, 514, 3 * 512 + 77, 8192 + 6):
        aos = torch.randint(-(1 << 62), 1 << 62, (n, N), dtype=torch.int64, device="cuda", generator=g)
        soa = F.from_aos(aos)
        assert torch.equal(soa, aos.t()), (P, n, "aos->soa")
        assert torch.equal(F.to_aos(soa), aos), (P, n, "soa->aos")
        wide = torch.zeros((N, n + 14), dtype=torch.int64, device="cuda")            # even padded stride: still the LDS kernels
        wide[:, :n] = soa
        assert torch.equal(F.to_aos(wide[:, :n]), aos)
        odd = torch.zeros((N, n + 13), dtype=torch.int64, device="cuda")             # odd stride: the plain kernels
        odd[:, :n] = soa
        assert torch.equal(F.to_aos(odd[:, :n]), aos)
    for tile, n in ((128, 5 * 128), (4096, 3 * 4096)):
        T = Field(P, tile=tile)
        aos = torch.randint(-(1 << 62), 1 << 62, (n, N), dtype=torch.int64, device="cuda", generator=g)
        tiled = T.from_aos(aos)
        assert tiled.dim() == 3 and torch.equal(tiled, tile_batch(aos.t().contiguous(), tile)), (P, tile)
        assert torch.equal(T.to_aos(tiled), aos)
