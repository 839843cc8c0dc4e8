#!/usr/bin/env python3
"""one fused ecn mul_get pass per curve (for profiling): ED25519 2^21, ED448 2^19, NIST256 / SECP256K1 2^20 scalars; with "gen" as first
argument one mulgen_get (2^21; ED448 2^20) and one mulgen2_get pass per curve instead"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
GEN = sys.argv[1:2] == ["gen"]
if GEN:
    del sys.argv[1]
    for name, n in (("ED25519", 1 << 21), ("ED448", 1 << 20), ("NIST256", 1 << 21), ("SECP256K1", 1 << 21)):
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        Ed = Curve(name)
        e = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda")
        Ed.mulgen_get(e[:4096].contiguous())
        x, y, s = Ed.mulgen_get(e)
        m = n // 2
        Q = Ed.mul(e[:m].contiguous(), Ed.gen(m))
        f = torch.randint(0, 256, (m, Ed.nbytes), dtype=torch.uint8, device="cuda")
        Ed.mulgen2_get(e[:4096].contiguous(), f[:4096].contiguous(), Q[:, :, :4096].contiguous())
        x, y, s = Ed.mulgen2_get(e[:m].contiguous(), f, Q)
        torch.cuda.synchronize()
    print("done")
    sys.exit(0)
for name, n in (("ED25519", 1 << 21), ("ED448", 1 << 19), ("NIST256", 1 << 20), ("SECP256K1", 1 << 20)):
    if sys.argv[1:] and name not in sys.argv[1:]:
        continue
    Ed = Curve(name)
    e = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda")
    k = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda")
    P = Ed.mul(k, Ed.gen(n))
    Ed.mul_get(e[:4096].contiguous(), P[:, :, :4096].contiguous())
    x, y, s = Ed.mul_get(e, P)
    torch.cuda.synchronize()
print("done")
