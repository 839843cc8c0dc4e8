#!/usr/bin/env python3
"""time series of the headline modmul: windows of 25 launches (8 ms), (i) 160 windows on one operand triple, (ii) alternating between
two triples, (iii) with a moduniform (VALU-heavy generator kernel over 2^24 elements) before every 4th window, (iv) with a 640 MiB
allocation + free before every 4th window -- what precedes the slow windows? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", torch.device("cuda", 0), tile=4096)
n = 1 << 24
T = [(F.uniform(n, array=0), F.uniform(n, array=1), F.empty(n)) for _ in range(2)]
scratch = F.empty(n)
torch.cuda.synchronize()


def window(t, k=25):
    a, b, c = t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120 * n * k / (e0.elapsed_time(e1) * 1e-3) / 1e9


def show(label, rates):
    slow = [i for i, r in enumerate(rates) if r < 6350]
    print("%-46s median %.0f  min %.0f  slow windows (< 6350): %d of %d at %s" % (label, sorted(rates)[len(rates) // 2], min(rates), len(slow), len(rates), slow[:24]), flush=True)


window(T[0]); window(T[1])
show("one triple, 160 windows", [window(T[0]) for _ in range(160)])
show("alternating two triples", [window(T[i & 1]) for i in range(160)])
r = []
for i in range(160):
    if i % 4 == 0:
        F.uniform(n, array=5, out=scratch)
    r.append(window(T[0]))
show("moduniform before every 4th window", r)
r = []
for i in range(160):
    if i % 4 == 0:
        x = torch.empty(5 * n, dtype=torch.int64, device="cuda"); del x; torch.cuda.empty_cache()
    r.append(window(T[0]))
show("alloc + free 640 MiB before every 4th window", r)
r = []
for i in range(160):
    if i % 4 == 0:
        scratch.copy_(T[1][0])
    r.append(window(T[0]))
show("640 MiB device copy before every 4th window", r)
