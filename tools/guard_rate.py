"""ecn mul / mul2 of four curves, event-timed with the shader-clock probe (round 6: what the limb-contract guard costs -- tools/prof_guard.sh
runs it under rocprofv3 --stats for the per-kernel durations of the fast and the exact launches)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
from modarith_amd.clock import timed_with_clock
g = torch.Generator(device="cuda").manual_seed(1)
for name, lg in (("ED25519", 20), ("ED448", 19), ("NIST256", 19), ("SECP256K1", 19)):
    C = Curve(name); n = 1 << lg
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    f = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    G = C.gen(n)
    t, ghz, Q = timed_with_clock(lambda: C.mul(e, G.clone()), reps=5, warm=2)
    t2, ghz2, _ = timed_with_clock(lambda: C.mul2(e, G, f, Q), reps=5, warm=2)
    print("%-10s ecn mul %.3e/s at %.2f GHz   mul2 %.3e/s at %.2f GHz   (two launches each: fast class + exact class behind it)" % (name, n / t, ghz or 0, n / t2, ghz2 or 0), flush=True)
