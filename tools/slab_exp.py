#!/usr/bin/env python3
"""placement experiment 5 (tiles of 4096): K fresh slabs, the operand triple carved out of each at 642 MiB distances, measured in two
role assignments; against K separately allocated triples in the same process (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", torch.device("cuda", 0), tile=4096)
n = 1 << 24
words = 5 * n
MiB = 1 << 20
K = int(os.environ.get("K", "10"))
src_a, src_b = F.uniform(n, array=0), F.uniform(n, array=1)


def rate(a, b, c):
    for _ in range(3):
        F.modmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


keep, rs, rr = [], [], []
for k in range(K):
    slab = torch.empty(3 * 642 * MiB // 8, dtype=torch.int64, device="cuda")
    at = lambda m: slab[m * MiB // 8: m * MiB // 8 + words].view(n // 4096, 5, 4096)
    a, b, c = at(1284), at(642), at(0)
    a.copy_(src_a); b.copy_(src_b)
    rs.append(rate(a, b, c)); rr.append(rate(b, c, a))
    keep.append(slab)
print("slabs    : a,b->c " + " ".join("%.0f" % r for r in rs))
print("           b,c->a " + " ".join("%.0f" % r for r in rr), flush=True)
rs, rr = [], []
for k in range(K):
    a, b, c = F.empty(n), F.empty(n), F.empty(n)
    a.copy_(src_a); b.copy_(src_b)
    rs.append(rate(a, b, c)); rr.append(rate(b, c, a))
    keep.append((a, b, c))
print("separate : a,b->c " + " ".join("%.0f" % r for r in rs))
print("           b,c->a " + " ".join("%.0f" % r for r in rr), flush=True)
