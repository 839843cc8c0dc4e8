#!/usr/bin/env python3
"""time ecn mul / mul2 for the built curves (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
only = sys.argv[1:]
for name, n in (("ED25519", 1 << 20), ("NIST256", 1 << 19), ("ED448", 1 << 18), ("NIST384", 1 << 18), ("NIST521", 1 << 17), ("SECP256K1", 1 << 19), ("NUMS256W", 1 << 19), ("NUMS256E", 1 << 19), ("ED248", 1 << 19), ("ED376", 1 << 18), ("ED500", 1 << 17)):
    if only and name not in only:
        continue
    Ed = Curve(name)
    e = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda")
    f = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda")
    P = Ed.gen(n)
    Ed.mul(e[:4096].contiguous(), P[:, :, :4096].contiguous()); torch.cuda.synchronize()
    t0 = time.perf_counter(); Ed.mul(e, P); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, n, "ecn mul  %.1f ms  %.3e per s" % (dt * 1e3, n / dt), flush=True)
    m = n // 4
    Pm, Qm = P[:, :, :m].contiguous(), Ed.gen(m)
    t0 = time.perf_counter(); Ed.mul2(e[:m].contiguous(), Pm, f[:m].contiguous(), Qm); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, m, "ecn mul2 %.1f ms  %.3e per s" % (dt * 1e3, m / dt), flush=True)
