#!/usr/bin/env python3
"""ONE pass of every VALU-bound leg of bench.py at a fixed size (tools/valu_legs.py), each kernel family launched exactly once
per curve -- the target of the counter passes of tools/gpu_valu_legs_pmc.sh.  With `--time`: no profiler, every leg event-timed
(median of 3 after 2 warm passes) with the shader-clock probe beside it."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from valu_legs import LOG2, records
from modarith_amd.field import rfc7748
from modarith_amd.edwards import Curve

TIME = "--time" in sys.argv
only = [a for a in sys.argv[1:] if not a.startswith("--")]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
res = {}


def run(name, fn):
    if not TIME:
        out = fn()
        torch.cuda.synchronize()
        return out
    from modarith_amd.clock import timed_with_clock
    t, ghz, out = timed_with_clock(fn)
    res[name] = {"records": records(name), "per_s": records(name) / t, "ms": t * 1e3, "sclk_GHz": ghz}
    print("%-32s %.4e/s  %8.2f ms  sclk %.3f GHz" % (name, records(name) / t, t * 1e3, ghz or 0), flush=True)
    return out


for curve, nb in (("X25519", 32), ("X448", 56)):
    if only and curve not in only:
        continue
    n = 1 << LOG2[curve.lower()]
    k = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device=dev, generator=g)
    u = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device=dev, generator=g)
    run(curve.lower(), lambda: rfc7748(curve, k, u))
for name in ("ED25519", "ED448", "NIST256", "SECP256K1"):
    if only and name not in only:
        continue
    Cv = Curve(name, dev)
    n = 1 << LOG2[name]
    m = n // 2
    rnd = lambda r: torch.randint(0, 256, (r, Cv.nbytes), dtype=torch.uint8, device=dev, generator=g)
    e, f = rnd(n), rnd(n)
    G = Cv.gen(n)
    Q = run(name + "_ecn_mul", lambda: Cv.mul(e, G.clone()))
    run(name + "_ecn_mul2", lambda: Cv.mul2(e, G, f, Q))
    run(name + "_ecn_mul_get_fused", lambda: Cv.mul_get(f, Q))
    e2, f2, G2, Q2 = e[:m].contiguous(), f[:m].contiguous(), G[:, :, :m].contiguous(), Q[:, :, :m].contiguous()
    run(name + "_ecn_mul2_get_fused", lambda: Cv.mul2_get(e2, G2, f2, Q2))
    run(name + "_ecn_mulgen_get_fused", lambda: Cv.mulgen_get(e))
    run(name + "_ecn_mulgen2_get_fused", lambda: Cv.mulgen2_get(e2, f2, Q2))
    del e, f, G, Q, e2, f2, G2, Q2
if TIME:
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/valu_leg_rates.json", "w"), indent=1)
print("done")
