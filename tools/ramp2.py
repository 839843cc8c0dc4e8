#!/usr/bin/env python3
"""per-launch duration of the headline modmul (tiles of 4096, 2^24 elements) (i) from the first launch of a process, (ii) after idle
pauses, (iii) after freeing 4 GiB of device memory -- how long does the chip take to reach its streaming rate again? (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", torch.device("cuda", 0), tile=4096)
n = 1 << 24
a, b = F.uniform(n, array=0), F.uniform(n, array=1)
c = torch.empty_like(a)
torch.cuda.synchronize()


def burst(K, label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for i in range(K):
        F.modmul(a, b, out=c)
        ev[i + 1].record()
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    print("%-34s first 10: %s | 11-30 mean %.0f | 31-60 mean %.0f | last 20 mean %.0f" % (label, " ".join("%.0f" % u for u in us[:10]), sum(us[10:30]) / 20, sum(us[30:60]) / 30, sum(us[-20:]) / 20), flush=True)


burst(120, "first launches of the process")
burst(120, "immediately again")
for pause in (0.005, 0.05, 0.5, 2.0):
    time.sleep(pause)
    burst(120, "after %.3f s idle" % pause)
junk = [torch.empty(1 << 27, dtype=torch.int64, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
del junk
torch.cuda.empty_cache()
burst(120, "after freeing 4 GiB (empty_cache)")
x = F.uniform(n, array=2)
burst(120, "after a moduniform + allocation")
