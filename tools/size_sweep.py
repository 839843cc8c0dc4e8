#!/usr/bin/env python3
"""streaming rate of the headline modmul against batch size, 2^20 .. 2^27 elements (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=None)     # flat rows: what this script measures and labels (Field() alone is tiled since round 4)
for lg in (20, 22, 24, 25, 26, 27):
    n = 1 << lg
    a = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
    b = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
    c = torch.empty_like(a)
    reps = max(4, (1 << 28) // n)
    for _ in range(3): F.modmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("2^%d elements: %.3f ms  %.0f GB/s" % (lg, ms, 120.0 * n / ms / 1e6), flush=True)
    del a, b, c
    torch.cuda.empty_cache()
