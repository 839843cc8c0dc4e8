"""GPU parity, round 3: the reference's corner-case protocol (edge.py / edge.c) through the HIP library for every built prime."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CORE = ["X25519", "NIST256", "X448"]
EXTRA = list(__import__("modarith_amd.emit", fromlist=["EXTRA_PRIMES"]).EXTRA_PRIMES)


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
    from modarith_amd.params import derive
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
