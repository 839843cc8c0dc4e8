#!/usr/bin/env python3
"""ecn mul2: the default constant-time form against the reference-exact walk (Curve.mul2(exact=True)), 2^18 pairs (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
n = 1 << 18
for name in ("ED25519", "SECP256K1", "NIST256", "ED448", "NIST384"):
    W = Curve(name)
    g = torch.Generator(device="cuda").manual_seed(3)
    e = torch.randint(0, 256, (n, W.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    f = torch.randint(0, 256, (n, W.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    P = W.mul(e.clone(), W.gen(n)); Q = W.dbl(P.clone())
    res = {}
    for exact in (False, True):
        W.mul2(e, P, f, Q, exact=exact); torch.cuda.synchronize()
        t0 = time.perf_counter(); R = W.mul2(e, P, f, Q, exact=exact); torch.cuda.synchronize()
        res[exact] = (n / (time.perf_counter() - t0), R)
    same = bool(W.cmp(res[False][1], res[True][1]).all())
    print("%-10s mul2 %.3g/s   mul2 exact %.3g/s   (x%.2f)   same points: %s" % (name, res[False][0], res[True][0], res[False][0] / res[True][0], same))
