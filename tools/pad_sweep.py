#!/usr/bin/env python3
"""does a padded limb stride (ld = n + pad) help at batch sizes where the rows sit a large power of two apart? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=None)     # flat rows: what this script measures and labels (Field() alone is tiled since round 4)
for lg in (int(x) for x in (sys.argv[1:] or ["25", "26"])):
    n = 1 << lg
    for pad in (0, 32, 544, 2112, 8224, 65568, 1 << 20):
        ld = n + pad
        bufs = [torch.randint(0, 1 << 51, (5, ld), dtype=torch.int64, device="cuda") for _ in range(3)]
        a, b, c = (t[:, :n] for t in bufs)
        reps = max(4, (1 << 28) // n)
        for _ in range(3): F.modmul(a, b, out=c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): F.modmul(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("2^%d elements, ld = n + %-8d %.3f ms  %.0f GB/s" % (lg, pad, ms, 120.0 * n / ms / 1e6), flush=True)
        del bufs, a, b, c
        torch.cuda.empty_cache()
