#!/usr/bin/env python3
"""Time the element-wise kernels for every built prime with HIP events (GPU box only).
   python tools/sweep.py [log2_n] [reps]      env: MA_MAX_BLOCKS, MA_FORCE_EPT1"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modarith_amd.field import Field

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 1 << log2n
res = {"n": n, "MA_MAX_BLOCKS": os.environ.get("MA_MAX_BLOCKS"), "MA_FORCE_EPT1": os.environ.get("MA_FORCE_EPT1")}
for P in ("X25519", "NIST256", "X448"):
    F = Field(P)
    N, radix = F.N, F.radix
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randint(0, 1 << radix, (N, n), dtype=torch.int64, device="cuda", generator=g)
    b = torch.randint(0, 1 << radix, (N, n), dtype=torch.int64, device="cuda", generator=g)
    c = torch.empty_like(a)
    if os.environ.get("MA_FORCE_EPT1"):
        # odd limb stride defeats the 16-byte path
        big = torch.empty((N, n + 1), dtype=torch.int64, device="cuda")
        a2, b2, c2 = big.clone(), big.clone(), big.clone()
        a2[:, :n] = a; b2[:, :n] = b
        a, b, c = a2[:, :n], b2[:, :n], c2[:, :n]
    ops = {
        "modmul": (lambda: F.modmul(a, b, out=c), 3),
        "modsqr": (lambda: F.modsqr(a, out=c), 2),
        "modadd": (lambda: F.modadd(a, b, out=c), 3),
        "modsub": (lambda: F.modsub(a, b, out=c), 3),
        "modmli": (lambda: F.modmli(a, 121665, out=c), 2),
        "redc": (lambda: F.redc(a, out=c), 2),
        "nres": (lambda: F.nres(a, out=c), 2),
        "modcpy": (lambda: F.modcpy(a, out=c), 2),
        "modmuls": (lambda: F.modmuls(a, [3, 1, 4, 1, 5, 9, 2, 6][:N], out=c), 2),
    }
    for name, (fn, arrays) in ops.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = arrays * 8 * N * n / (ms * 1e-3) / 1e9
        res["%s_%s" % (P, name)] = {"ms": round(ms, 4), "GBps": round(gbs, 1), "Gops": round(n / ms / 1e6, 2)}
print(json.dumps(res))
