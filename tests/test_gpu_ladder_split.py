"""The split form of the batched ladders (csrc/fe_finish.h): ladders in one kernel, then one inversion per up to 32 records
(Montgomery's simultaneous inversion) in a second one.  Its bytes must equal the one-inversion-per-record kernel's and the
oracle's for every input -- in particular for records whose z2 is zero (u = 0 and the low-order points of RFC 7748 section 7),
which would annihilate a whole group's product if they were not taken out by lane predication."""
import ctypes
import os

import numpy as np
import pytest

from tests.oracle_binding import PRIMES
from tests.util import vp

pytestmark = pytest.mark.gpu

P25519 = (1 << 255) - 19
P448 = (1 << 448) - (1 << 224) - 1
# points of small order on Curve25519 / its twist (RFC 7748 section 7 refers to them; the list is the usual one) and on Curve448
LOW = {
    "X25519": [0, 1, 325606250916557431795983626356110631294008115727848805560023387167927233504,
               39382357235489614581723060781553021112529911719440698176882885853963445705823, P25519 - 1, P25519, P25519 + 1,
               2 * P25519 - 1 if 2 * P25519 - 1 < (1 << 255) else 0],
    "X448": [0, 1, P448 - 1, P448, P448 + 1],
}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _inputs(C, n, seed):
    nb = PRIMES[C][3]
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    u = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    low = [np.frombuffer(int(v % (1 << (8 * nb))).to_bytes(nb, "little"), dtype=np.uint8) for v in LOW[C]]
    for i, j in enumerate(range(0, n, 997)):            # sprinkled through the batch: every group of the second kernel meets some
        u[j] = low[i % len(low)]
    u[1:1 + len(low)] = np.stack(low)                   # and a run of consecutive ones
    u[n - 1] = low[0]
    return k, u


def _one_per_record(lib, C, k, u):
    """the self-contained kernel: rfc7748_<C>_batch in slices below the split threshold"""
    import torch
    out = torch.empty_like(u)
    f = getattr(lib, "rfc7748_%s_batch" % C)
    n = k.shape[0]
    for lo in range(0, n, 4096):
        hi = min(n, lo + 4096)
        assert f(k[lo:hi].data_ptr(), u[lo:hi].data_ptr(), out[lo:hi].data_ptr(), hi - lo, None) == 0
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("C,n", [("X25519", 8192), ("X25519", 65537), ("X25519", 4 * 65536 + 77), ("X25519", 33 * 65536 + 3),
                                 ("X448", 8192 + 1), ("X448", 2 * 65536 + 5)])
def test_split_equals_one_inversion_per_record(oracle, torch_cuda, C, n):
    torch = torch_cuda
    from modarith_amd import _lib
    from modarith_amd.field import rfc7748
    lib = _lib.load()
    k, u = _inputs(C, n, 1000 + n % 1000)
    K, U = torch.from_numpy(k).cuda(), torch.from_numpy(u).cuda()
    want = _one_per_record(lib, C, K, U)
    got_ws = rfc7748(C, K, U)                                            # caller-supplied workspace (torch) -> _batch_ws
    assert torch.equal(got_ws, want), "split form (batch_ws) differs from the one-inversion-per-record kernel"
    got_pool = torch.empty_like(U)                                       # the plain entry point: the library's own pool
    assert getattr(lib, "rfc7748_%s_batch" % C)(K.data_ptr(), U.data_ptr(), got_pool.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(got_pool, want), "split form (own pool) differs"
    # aliasing bv == bu (rfc7748.c:329) in the split form
    U2 = U.clone()
    rfc7748(C, K, U2, out=U2)
    assert torch.equal(U2, want)
    # low-order inputs give the all-zero record, as in the reference (x2 * 0^(p-2))
    w = want.cpu().numpy()
    assert not w[1].any() and not w[n - 1].any()
    # a sample against the oracle: the sprinkled records, their neighbours and a stride
    idx = np.unique(np.concatenate([np.arange(0, n, 997), np.arange(0, min(n, 64)), np.arange(n - 64, n), np.arange(0, n, max(n // 3000, 1))]))
    ks, us = np.ascontiguousarray(k[idx]), np.ascontiguousarray(u[idx])
    ref = np.empty_like(us)
    oracle.lib.oracle_parallel(3 if C == "X25519" else 4, vp(ks), vp(us), vp(ref), len(idx), 0, len(os.sched_getaffinity(0)))
    assert np.array_equal(w[idx], ref), "split form differs from the oracle"


def test_split_workspace_contract(torch_cuda):
    torch = torch_cuda
    from modarith_amd import _lib
    lib = _lib.load()
    n = 10000
    assert lib.rfc7748_X25519_batch_workspace_bytes(n) == n * (32 + 40)
    assert lib.rfc7748_X448_batch_workspace_bytes(n) == n * (56 + 64)
    k = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    ws = torch.empty(n * 72 // 8, dtype=torch.int64, device="cuda")
    assert lib.rfc7748_X25519_batch_ws(k.data_ptr(), k.data_ptr(), k.data_ptr(), n, ws.data_ptr(), n * 72 - 8, None) != 0
    assert b"workspace" in lib.modarith_amd_last_error()
    assert lib.rfc7748_X25519_batch_ws(k.data_ptr(), k.data_ptr(), k.data_ptr(), n, None, 1 << 30, None) != 0
    # small n is legal for the _ws entry (rounds = 1)
    g = torch.Generator(device="cuda").manual_seed(3)
    kk = torch.randint(0, 256, (77, 32), dtype=torch.uint8, device="cuda", generator=g)
    uu = torch.randint(0, 256, (77, 32), dtype=torch.uint8, device="cuda", generator=g)
    a, b = torch.empty_like(uu), torch.empty_like(uu)
    assert lib.rfc7748_X25519_batch_ws(kk.data_ptr(), uu.data_ptr(), a.data_ptr(), 77, ws.data_ptr(), ws.numel() * 8, None) == 0
    assert lib.rfc7748_X25519_batch(kk.data_ptr(), uu.data_ptr(), b.data_ptr(), 77, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_split_under_graph_capture(torch_cuda):
    """captured on a side stream: the Python wrapper's workspace comes from torch's graph-aware allocator (split form inside
    the graph); the plain C entry point sees the capture and enqueues the self-contained kernel instead of allocating"""
    torch = torch_cuda
    from modarith_amd import _lib
    from modarith_amd.field import rfc7748
    lib = _lib.load()
    n = 3 * 8192 + 1
    g = torch.Generator(device="cuda").manual_seed(11)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    u = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    out1, out2 = torch.empty_like(u), torch.empty_like(u)
    want = rfc7748("X25519", k, u)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rfc7748("X25519", k, u, out=out1)
            assert lib.rfc7748_X25519_batch(k.data_ptr(), u.data_ptr(), out2.data_ptr(), n, ctypes.c_void_p(side.cuda_stream)) == 0
    out1.zero_(); out2.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out1, want) and torch.equal(out2, want)
