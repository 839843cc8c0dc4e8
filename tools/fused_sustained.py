#!/usr/bin/env python3
"""The fused Weierstrass kernels over time (round 5: three waves per SIMD, multiplier-dense): one launch per sample (2^20 records), every
launch timed with HIP events, for `secs` seconds per kernel from an idle part; the rate of the first launches and the mean over successive
one-second windows -- is the bench figure a cold reading or the steady state?
    python tools/fused_sustained.py [secs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
m = 1 << 20
for name in ("NIST256", "SECP256K1"):
    C = Curve(name)
    rnd = lambda: torch.randint(0, 256, (m, C.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    e, f = rnd(), rnd()
    P = C.mul(rnd(), C.gen(m))
    for leg, call in (("mul_get", lambda: C.mul_get(e, P)), ("mulgen2_get", lambda: C.mulgen2_get(e, f, P)), ("mulgen_get", lambda: C.mulgen_get(e))):
        call()
        torch.cuda.synchronize()
        time.sleep(2.0)
        ev, t0 = [], time.time()
        while time.time() - t0 < secs:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); call(); b.record()
            ev.append((a, b))
            if len(ev) % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in ev]
        line = "%-10s %-12s %4d launches of 2^20; first three %s; per second:" % (name, leg, len(ms), " ".join("%.3e" % (m / (x * 1e-3)) for x in ms[:3]))
        acc, k = 0.0, 0
        for x in ms:
            acc += x; k += 1
            if acc >= 1000.0:
                line += " %.3e" % (m * k / (acc * 1e-3))
                acc, k = 0.0, 0
        print(line, flush=True)
