#!/usr/bin/env python3
"""one-off: 2^27 elements per array (5.4 GB each, element offsets beyond 2^32 bytes) through the streaming kernels:
commutativity, a strided sample against the oracle, and the rate (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from modarith_amd.field import Field
from tests.oracle_binding import load_oracle
from tests.util import vp
F = Field("X25519", tile=None); n = 1 << 27     # flat rows (Field() alone is tiled since round 4)
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda", generator=g)
b = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda", generator=g)
c = F.modmul(a, b); torch.cuda.synchronize()
t0 = time.perf_counter(); F.modmul(a, b, out=c); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("2^27 modmul: %.2f ms, %.0f GB/s" % (dt * 1e3, 120 * n / dt / 1e9))
d = F.modmul(b, a)
print("commutative bit for bit:", bool(torch.equal(c, d)))
idx = torch.cat([torch.arange(0, 4096), torch.arange(n - 4096, n), torch.arange(0, n, n // 8192)]).unique().cuda()
o = load_oracle(build=not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so")))
ha = np.ascontiguousarray(a[:, idx].cpu().numpy().view(np.uint64)); hb = np.ascontiguousarray(b[:, idx].cpu().numpy().view(np.uint64)); hc = np.empty_like(ha)
o.fn("batch_modmul", "X25519")(vp(ha), vp(hb), vp(hc), ha.shape[1], ha.shape[1])
print("sample of %d elements equals the oracle:" % idx.numel(), bool(np.array_equal(c[:, idx].cpu().numpy().view(np.uint64), hc)))
