#!/usr/bin/env python3
"""in-register chains (the reference's time.c protocol per lane, k_time): modmul / modsqr per second when operands
never leave the VGPRs -- the VALU ceiling of the field arithmetic, next to the HBM-bound streaming rate (GPU box).
Run plain (exact 128-bit products) and with MA_FORCE_FAST=1 (split products on the 64-bit column chain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
from modarith_amd.params import derive
n = 1 << 20
for name in sys.argv[1:] or ["X25519", "NIST256", "X448"]:
    fp = derive(name); F = Field(name)
    x = torch.randint(0, 1 << fp.radix, (fp.nlimbs, n), dtype=torch.int64, device="cuda")
    x[fp.nlimbs - 1] &= (1 << (fp.n - fp.radix * (fp.nlimbs - 1) - 1)) - 1
    y = x.flip(1).contiguous()
    for kind, per in (("modmul", 1000), ("modsqr", 1000)):
        F.time_protocol(kind, x[:, :4096].contiguous(), y[:, :4096].contiguous() if kind == "modmul" else None, 1); torch.cuda.synchronize()
        t0 = time.perf_counter(); F.time_protocol(kind, x, y if kind == "modmul" else None, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-8s %-7s %7.2f ms for 2^20 lanes x %d  = %.3e per s" % (name, kind, dt * 1e3, per, n * per / dt), flush=True)
