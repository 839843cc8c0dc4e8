#!/usr/bin/env python3
"""placement experiment 3: operand triples allocated exactly as bench.py allocates them (F.uniform, F.uniform, empty_like), their
device addresses and their rates -- looking for what the slow placements have in common (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", torch.device("cuda", 0), tile=4096)
n = 1 << 24
K = int(os.environ.get("K", "12"))


def rate(a, b, c):
    for _ in range(3):
        F.modmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


if os.environ.get("WARM") == "1":
    # initialise the engine before allocating the operands: the first launch loads the library's code objects (a few hundred MiB of
    # device memory), which otherwise lands BETWEEN the first operand and the second
    w = F.uniform(4096, seed=1, array=9)
    F.modmul(w, w)
    torch.cuda.synchronize()
keep = []
for k in range(K):
    a = F.uniform(n, seed=42, array=0); b = F.uniform(n, seed=42, array=1); c = torch.empty_like(a)
    r = rate(a, b, c)
    pa, pb, pc = a.data_ptr(), b.data_ptr(), c.data_ptr()
    # roles permuted on the same memory: is it the memory or the role?
    r2 = rate(b, c, a)
    print("triple %2d: %.0f GB/s (roles rotated: %.0f)   a %#x  b-a %+d MiB  c-b %+d MiB   a mod 4 GiB = %d MiB" % (k, r, r2, pa, (pb - pa) >> 20, (pc - pb) >> 20, (pa % (1 << 32)) >> 20), flush=True)
    keep.append((a, b, c))
