#!/usr/bin/env python3
"""placement experiment 4: the three operand arrays of the headline modmul carved out of ONE slab at controlled byte distances
(a at A, b at B, c at C MiB): which distances stream at the full rate? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", torch.device("cuda", 0), tile=4096)
n = 1 << 24
words = 5 * n
MiB = 1 << 20
slab = torch.empty(7 * 1024 * MiB // 8, dtype=torch.int64, device="cuda")
src_a, src_b = F.uniform(n, array=0), F.uniform(n, array=1)
print("slab at %#x" % slab.data_ptr())


def at(mib):
    o = mib * MiB // 8
    return slab[o:o + words].view(n // 4096, 5, 4096)


def rate(A, B, C):
    a, b, c = at(A), at(B), at(C)
    a.copy_(src_a); b.copy_(src_b)
    for _ in range(3):
        F.modmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


print("equal distances s (a = 2s, b = s, c = 0: descending, as hipMalloc hands them out):")
for s in (640, 641, 642, 643, 644, 646, 648, 656, 672, 686, 704, 736, 768, 800, 832, 896, 960, 1024, 1070, 1072, 1088, 1280):
    print("  s = %4d MiB: %.0f GB/s" % (s, rate(2 * s, s, 0)), flush=True)
print("the slow triple of placement3 (b - a = -1070, c - b = -642) and neighbours:")
for da, db in ((1070, 642), (1072, 642), (1068, 642), (1066, 642), (1074, 642), (1038, 642), (1102, 642), (642, 1070), (1070, 1070), (856, 642), (1284, 642)):
    print("  a - b = %4d, b - c = %4d MiB: %.0f GB/s" % (da, db, rate(db + da, db, 0)), flush=True)
