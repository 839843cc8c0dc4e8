#!/usr/bin/env python3
"""flat [N, n] batches against the same elements laid out as tiles [n/tile][N][tile] (one batched call per tile,
ld = tile): does the 2^25-2^26 bandwidth dip of the flat layout come from how many distant pages a wave touches?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=None)     # flat rows: what this script measures and labels (Field() alone is tiled since round 4)
def rate(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for lg in (24, 25, 26, 27):
    n = 1 << lg
    a = F.uniform(n, array=0); b = F.uniform(n, array=1); c = torch.empty_like(a)
    ms = rate(lambda: F.modmul(a, b, out=c))
    line = "2^%d flat %.0f GB/s" % (lg, 120 * n / ms / 1e6)
    del a, b, c
    for tl in (20, 22):
        tile = 1 << tl
        T = n // tile
        ta = torch.empty((T, 5, tile), dtype=torch.int64, device="cuda"); tb = torch.empty_like(ta); tc = torch.empty_like(ta)
        for i in range(T):
            F.uniform(tile, array=0, first=i * tile, out=ta[i]); F.uniform(tile, array=1, first=i * tile, out=tb[i])
        def run():
            for i in range(T):
                F.modmul(ta[i], tb[i], out=tc[i])
        ms = rate(run)
        line += "   tiles of 2^%d: %.0f GB/s" % (tl, 120 * n / ms / 1e6)
        del ta, tb, tc
    print(line, flush=True)
