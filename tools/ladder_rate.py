#!/usr/bin/env python3
"""X25519 / X448 batched ladder rate on the GPU box (2^23 / 2^21 scalars, best of 3) + oracle check of a sample.
   usage: ladder_rate.py [X25519|X448 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from modarith_amd.field import rfc7748, rfc7748_base
from tests.oracle_binding import load_oracle
from tests.util import vp
oracle = load_oracle(build=False)
for curve in (sys.argv[1:] or ["X25519", "X448"]):
    nb, n = (32, 1 << 23) if curve == "X25519" else (56, 1 << 21)
    g = torch.Generator(device="cuda").manual_seed(5)
    k = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda", generator=g)
    u = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda", generator=g)
    o = torch.empty_like(u)
    rfc7748(curve, k[:8192], u[:8192], out=o[:8192]); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); rfc7748(curve, k, u, out=o); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    m = 4096
    hk, hu = np.ascontiguousarray(k[:m].cpu().numpy()), np.ascontiguousarray(u[:m].cpu().numpy())
    want = np.empty_like(hu)
    oracle.lib.oracle_parallel(3 if curve == "X25519" else 4, vp(hk), vp(hu), vp(want), m, 0, 32)
    ok = np.array_equal(o[:m].cpu().numpy(), want)
    print("%s: %.4g scalar mults/s (%.2f ms for 2^%d), first %d vs oracle: %s" % (curve, n / best, best * 1e3, n.bit_length() - 1, m, "EQUAL" if ok else "MISMATCH"), flush=True)

    # public-key generation: the same function on the base point, ladder against the fixed-base kernel
    ub = torch.zeros_like(u); ub[:, 0] = 9 if curve == "X25519" else 5
    rfc7748_base(curve, k[:8192].contiguous()); torch.cuda.synchronize()
    bl = bb = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); wl = rfc7748(curve, k, ub); torch.cuda.synchronize(); bl = min(bl, time.perf_counter() - t0)
        t0 = time.perf_counter(); wb = rfc7748_base(curve, k); torch.cuda.synchronize(); bb = min(bb, time.perf_counter() - t0)
    print("%s base point: fixed-base kernel %.4g/s (%.2f ms), ladder %.4g/s, ratio %.2f, equal: %s" % (curve, n / bb, bb * 1e3, n / bl, bl / bb, bool(torch.equal(wl, wb))), flush=True)
