#!/usr/bin/env python3
"""ecn mul over time: is the rate of the multiplier-dense scalar multiplication kernels a cold-start reading, a steady state, or
does it sag under sustained load?  One kernel launch per sample (2^20 points, 2^19 for ED448), every launch timed with HIP events,
for `secs` seconds per curve; prints the rate of the first launches and the mean over successive one-second windows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
for name, m in (("ED25519", 1 << 20), ("ED448", 1 << 19), ("NIST256", 1 << 19)):
    C = Curve(name)
    e = torch.randint(0, 256, (m, C.nbytes), dtype=torch.uint8, device="cuda", generator=gen)
    P = C.gen(m)
    C.mul(e, P)                                     # loads the code object, allocates the workspace
    torch.cuda.synchronize()
    time.sleep(2.0)                                 # start from an idle part
    ev, t0 = [], time.time()
    while time.time() - t0 < secs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); C.mul(e, P); b.record()
        ev.append((a, b))
        if len(ev) % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    rate = [m / (x * 1e-3) for x in ms]
    print("%s: %d launches of %d points; first five %s" % (name, len(ms), m, " ".join("%.3e" % r for r in rate[:5])))
    acc, k, win = 0.0, 0, 1
    for i, x in enumerate(ms):
        acc += x; k += 1
        if acc >= 1000.0:
            print("   second %d: %.3e/s (%d launches)" % (win, m * k / (acc * 1e-3), k))
            acc, k, win = 0.0, 0, win + 1
