#!/usr/bin/env python3
"""throughput of the VALU-bound batched field kernels (modinv, modsqrt, modpro, modqr) per prime (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
from modarith_amd.params import derive
primes = sys.argv[1:] or ["X25519", "NIST256", "X448"]
n = 1 << 20
for name in primes:
    fp = derive(name)
    F = Field(name)
    a = torch.randint(0, 1 << fp.radix, (fp.nlimbs, n), dtype=torch.int64, device="cuda")
    a[fp.nlimbs - 1] &= (1 << (fp.n - fp.radix * (fp.nlimbs - 1))) - 1
    for op in ("modinv", "modsqrt", "modpro"):
        fn = getattr(F, op)
        fn(a[:, :4096].contiguous()); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(a); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-9s %-8s %7.2f ms  %.3e per s" % (name, op, dt * 1e3, n / dt), flush=True)
