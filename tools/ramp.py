#!/usr/bin/env python3
"""per-launch duration of the first launches of the headline modmul in a fresh process (clock ramp)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519")
n = 1 << 24
a = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
b = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
c = torch.empty_like(a)
torch.cuda.synchronize()
K = 160
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
ev[0].record()
for i in range(K):
    F.modmul(a, b, out=c)
    ev[i + 1].record()
torch.cuda.synchronize()
print(" ".join("%.0f" % (ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(K)))
