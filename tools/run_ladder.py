#!/usr/bin/env python3
"""one X25519 and one X448 batched ladder pass (for profiling): 2^22 / 2^20 scalars"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import rfc7748
for curve, nb, n in (("X25519", 32, 1 << 22), ("X448", 56, 1 << 20)):
    k = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda")
    u = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda")
    rfc7748(curve, k[:4096].contiguous(), u[:4096].contiguous())
    out = rfc7748(curve, k, u)
    torch.cuda.synchronize()
print("done")
