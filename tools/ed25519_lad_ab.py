#!/usr/bin/env python3
"""round 5: the fused ED25519 kernels in their ladder form (csrc/ed26l.h) against the window form (MA_ED25519_FUSED=window), one
process per variant (the library reads its knobs once): rate of mul_get / mulgen2_get at 2^LOG2N records (median of 5 event-timed calls
after 2 warm calls), bytes compared with the two- / three-call forms.  tools/gpu_r05_lad_ab.sh runs the variants."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve

lg = int(os.environ.get("LOG2N", "20"))
n = 1 << lg
Ed = Curve("ED25519")
g = torch.Generator(device="cuda").manual_seed(3)
rnd = lambda m=n: torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda", generator=g)
e, f, k = rnd(), rnd(), rnd()
P = Ed.mul(k, Ed.gen(n))


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        out = fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return sorted(ts)[len(ts) // 2], out


tag = "window" if os.environ.get("MA_ED25519_FUSED") == "window" else "ladder"
t, (x, y, _) = timed(lambda: Ed.mul_get(e, P))
wx, wy, _ = Ed.get(Ed.mul(e, P.clone()))
dig = hashlib.sha256(x.cpu().numpy().tobytes() + y.cpu().numpy().tobytes()).hexdigest()[:16]
print("%-28s 2^%d mul_get     %.4e/s (%.2f ms)  equal to mul + get: %s  digest %s" % (tag, lg, n / t, t * 1e3, bool(torch.equal(x, wx) and torch.equal(y, wy)), dig), flush=True)
m = n // 2
e2, f2, Q2 = e[:m].contiguous(), f[:m].contiguous(), P[:, :, :m].contiguous()
t, (x, y, _) = timed(lambda: Ed.mulgen2_get(e2, f2, Q2))
wx, wy, _ = Ed.get(Ed.mul2(e2, Ed.gen(m), f2, Q2))
dig = hashlib.sha256(x.cpu().numpy().tobytes() + y.cpu().numpy().tobytes()).hexdigest()[:16]
print("%-28s 2^%d mulgen2_get %.4e/s (%.2f ms)  equal to gen + mul2 + get: %s  digest %s" % (tag, lg - 1, m / t, t * 1e3, bool(torch.equal(x, wx) and torch.equal(y, wy)), dig), flush=True)
