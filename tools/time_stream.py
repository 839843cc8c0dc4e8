#!/usr/bin/env python3
"""streaming rates of the unary / binary field kernels for one prime at 2^24 elements (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
from modarith_amd.params import derive
for name in sys.argv[1:] or ["X448", "NIST256"]:
    fp = derive(name); F = Field(name); n = 1 << 24
    a = torch.randint(0, 1 << fp.radix, (fp.nlimbs, n), dtype=torch.int64, device="cuda")
    b = torch.randint(0, 1 << fp.radix, (fp.nlimbs, n), dtype=torch.int64, device="cuda")
    c = torch.empty_like(a)
    for op, nb, fn in (("modmul", 3, lambda: F.modmul(a, b, out=c)), ("modsqr", 2, lambda: F.modsqr(a, out=c)),
                       ("nres", 2, lambda: F.nres(a, out=c)), ("redc", 2, lambda: F.redc(a, out=c)), ("modadd", 3, lambda: F.modadd(a, b, out=c))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-8s %-7s %.3f ms  %.0f GB/s" % (name, op, ms, nb * 8 * fp.nlimbs * n / ms / 1e6), flush=True)
