#!/usr/bin/env python3
"""placement experiment 2: operand triples allocated back to back with `extra` bytes added to every allocation (so that consecutive
arrays sit 640 MiB + extra apart when the allocator places them consecutively); K fresh triples per setting (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=4096)
n = 1 << 24
words = 5 * n
K = int(os.environ.get("K", "8"))
src_a, src_b = F.uniform(n, array=0), F.uniform(n, array=1)


def alloc(extra):
    buf = torch.empty(words + extra // 8, dtype=torch.int64, device="cuda")
    return buf, buf[:words].view(n // 4096, 5, 4096)


def rate(a, b, c):
    for _ in range(3):
        F.modmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


MiB = 1 << 20
for extra in [0, 2 * MiB, 4 * MiB, 6 * MiB, 10 * MiB, 34 * MiB, 0, 2 * MiB]:
    rates, keep, gaps = [], [], []
    for k in range(K):
        ba, a = alloc(extra); bb, b = alloc(extra); bc, c = alloc(extra)
        a.copy_(src_a); b.copy_(src_b)
        rates.append(rate(a, b, c))
        gaps.append(((bb.data_ptr() - ba.data_ptr()) / MiB, (bc.data_ptr() - bb.data_ptr()) / MiB))
        keep.append((ba, bb, bc))
    del keep
    torch.cuda.empty_cache()
    print("extra %3d MiB: min %.0f median %.0f max %.0f | %s | gaps(MiB) %s" % (extra // MiB, min(rates), sorted(rates)[len(rates) // 2], max(rates),
          " ".join("%.0f" % r for r in rates), " ".join("%.0f/%.0f" % g for g in gaps)), flush=True)
