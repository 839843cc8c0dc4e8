#!/usr/bin/env python3
"""One box, one process: modmul against its streaming controls modadd (same three streams, no multiplication) and modcpy;
X25519 / NIST256 / X448 at 2^24 elements, interleaved rounds, median of 7.  (Round 2 also ran a two-stage software-pipelined
variant of the binary kernel through this script -- next iteration's row loads in flight during the product, 148 VGPRs --
and measured no difference: X25519 5 564 vs 5 564 GB/s, NIST256 5 625 vs 5 613, X448 5 252 vs 5 309; the variant was
removed again)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
n = 1 << 24
for P in sys.argv[1:] or ["X25519", "NIST256", "X448"]:
    F = Field(P)
    a = F.uniform(n, array=0); b = F.uniform(n, array=1); c = torch.empty_like(a)
    def rate(fn, reps=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    res = {"modmul": [], "modcpy": [], "modadd": []}
    for _ in range(7):
        res["modmul"].append(rate(lambda: F.modmul(a, b, out=c)))
        res["modadd"].append(rate(lambda: F.modadd(a, b, out=c)))
        res["modcpy"].append(rate(lambda: F.modcpy(a, out=c)))
    for k, v in res.items():
        ms = statistics.median(v)
        nb = (2 if k == "modcpy" else 3) * 8 * F.N * n
        print("%-8s %-7s %.4f ms  %.0f GB/s   (min %.4f max %.4f)" % (P, k, ms, nb / ms / 1e6, min(v), max(v)), flush=True)
