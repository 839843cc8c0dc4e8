#!/usr/bin/env python3
"""throughput of the batched group-law kernels (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
for name in ("ED25519", "NIST256"):
    Ed = Curve(name)
    n = 1 << 21
    e = torch.randint(0, 256, (4096, Ed.nbytes), dtype=torch.uint8, device="cuda")
    base = Ed.mul(e, Ed.gen(4096))
    P = base.repeat(1, 1, n // 4096).contiguous()
    Q = Ed.dbl(P.clone())
    for op, fn in (("dbl", lambda: Ed.dbl(P)), ("add", lambda: Ed.add(Q, P))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(name, op, "%.2f ms  %.3e per s" % (dt * 1e3, n / dt), flush=True)
