#!/usr/bin/env python3
"""streaming rates of the remaining element-wise kernels at 2^24 elements (GPU box): modcsw / modcmv (per-element selector),
modneg, modmli, modfsb, modis0, modcmp, modshl -- flat and tiled"""
import os, sys
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
        a, b = F.uniform(n, array=1), F.uniform(n, array=2)
        d = (torch.arange(n, device="cuda") % 3 == 0).to(torch.int32)
        row = 8 * F.N * n
        out = {
            "modcsw": rate(lambda: F.modcsw(d, a, b), 4 * row + 4 * n),
            "modcmv": rate(lambda: F.modcmv(d, a, b), 3 * row + 4 * n),
            "modneg": rate(lambda: F.modneg(a, out=b), 2 * row),
            "modmli": rate(lambda: F.modmli(a, 121665, out=b), 2 * row),
            "modfsb": rate(lambda: F.modfsb(b), 2 * row + 4 * n),
            "modis0": rate(lambda: F.modis0(a), row + 4 * n),
            "modcmp": rate(lambda: F.modcmp(a, b), 2 * row + 4 * n),
            "modshl": rate(lambda: F.modshl(1, b), 2 * row),
        }
        print("%-8s %-10s " % (name, "tile 4096" if tile else "flat") + "  ".join("%s %4.0f" % kv for kv in out.items()), flush=True)
