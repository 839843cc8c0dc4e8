#!/usr/bin/env python3
"""per-window launch time of the headline modmul over ~3 s in one process: does the rate drift or switch modes?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519")
n = 1 << 24
a = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
b = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
c = torch.empty_like(a)
torch.cuda.synchronize()
out = []
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 100 * 1e3)
print(" ".join("%.0f" % v for v in out))
