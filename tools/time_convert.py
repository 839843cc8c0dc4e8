#!/usr/bin/env python3
"""rates of the layout converters (element-major AoS <-> limb-interleaved SoA, flat and tiled) and of the byte-record import /
export (modimp / modexp) at 2^24 elements (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
n = 1 << 24
def rate(fn, nbytes, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
for name in sys.argv[1:] or ["X25519", "X448"]:
    for tile in (None, 4096):
        F = Field(name, tile=tile)
        a = F.uniform(n)
        aos = F.to_aos(a)
        nb = 2 * 8 * F.N * n
        r1 = rate(lambda: F.to_aos(a), nb)
        r2 = rate(lambda: F.from_aos(aos), nb)
        by = F.modexp(a)
        r3 = rate(lambda: F.modexp(a), (8 * F.N + F.nbytes) * n)
        r4 = rate(lambda: F.modimp(by), (8 * F.N + F.nbytes + 4) * n)
        print("%-8s %-10s soa->aos %5.0f GB/s  aos->soa %5.0f GB/s  modexp %5.0f GB/s  modimp %5.0f GB/s" % (name, "tile 4096" if tile else "flat", r1, r2, r3, r4), flush=True)
