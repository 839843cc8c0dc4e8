#!/usr/bin/env python3
"""PCIe-inclusive rate of a host-resident batch (for DESIGN.md section 5; never the bench `value`):
pinned host SoA operands -> H2D, modmul, D2H of the result, 2^24 elements of 2^255-19."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519")
n = 1 << 24
ha = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64).pin_memory()
hb = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64).pin_memory()
hc = torch.empty((5, n), dtype=torch.int64).pin_memory()
da, db, dc = torch.empty_like(ha, device="cuda"), torch.empty_like(hb, device="cuda"), torch.empty_like(hc, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    da.copy_(ha, non_blocking=True); db.copy_(hb, non_blocking=True)
    F.modmul(da, db, out=dc)
    hc.copy_(dc, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("PCIe-inclusive: %.1f ms per 2^24 modmul = %.3e modmul/s (%.1f GB/s over the link)" % (dt * 1e3, n / dt, 120 * n / dt / 1e9))

# the same job through modarith_amd.hostio (three streams, two device slots, chunks of 2^21 elements, C ABI only)
from modarith_amd.hostio import PinnedArray, map_host
pa, pb, pc = PinnedArray(5, n), PinnedArray(5, n), PinnedArray(5, n)
pa.array[:] = ha.numpy().view("uint64"); pb.array[:] = hb.numpy().view("uint64")
for chunk in (1 << 17, 1 << 18, 1 << 19, 1 << 20):
    map_host("X25519", "modmul", pa, pb, pc, chunk=chunk)
    t0 = time.perf_counter(); map_host("X25519", "modmul", pa, pb, pc, chunk=chunk); dt = time.perf_counter() - t0
    ok = bool((pc.array == hc.numpy().view("uint64")).all())
    print("pipelined chunk=2^%d: %.1f ms per 2^24 modmul = %.3e modmul/s (%.1f GB/s both directions summed) equal=%s"
          % (chunk.bit_length() - 1, dt * 1e3, n / dt, 120 * n / dt / 1e9, ok))
