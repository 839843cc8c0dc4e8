#!/usr/bin/env python3
"""placement experiment: the headline kernel (X25519 modmul, 2^24 elements on tiles of 4096) on operand triples whose second and third
array start at a byte offset (skew, 2 x skew) from a 2 MiB-aligned allocation -- does de-correlating the three streams' low address
bits remove the slow placements?  K fresh triples per skew; GB/s at 120 B per element (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=4096)
n = 1 << 24
words = 5 * n
K = int(os.environ.get("K", "6"))
src_a, src_b = F.uniform(n, array=0), F.uniform(n, array=1)


def view(skew_bytes):
    buf = torch.empty(words + skew_bytes // 8 + 16, dtype=torch.int64, device="cuda")
    return buf, buf[skew_bytes // 8: skew_bytes // 8 + words].view(n // 4096, 5, 4096)


def rate(a, b, c):
    for _ in range(3):
        F.modmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


for skew in [0, 256, 4096, 65536, 69632, 266240, 1052672, 3 * 1048576 + 4096]:
    rates, keep = [], []
    for k in range(K):
        ba, a = view(0); bb, b = view(skew); bc, c = view(2 * skew)
        a.copy_(src_a); b.copy_(src_b)
        rates.append(rate(a, b, c))
        keep.append((ba, bb, bc))                       # hold the buffers so that every triple is a fresh placement
    del keep
    torch.cuda.empty_cache()
    print("skew %8d B: min %.0f  median %.0f  max %.0f   %s" % (skew, min(rates), sorted(rates)[len(rates) // 2], max(rates), " ".join("%.0f" % r for r in rates)), flush=True)
