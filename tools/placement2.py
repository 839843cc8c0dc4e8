#!/usr/bin/env python3
"""placement study, part 2 (GPU box): three operand arrays as views of ONE allocation (with small skews between them) against
three separate allocations, several of each alive at once, each timed twice in alternating order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=None)     # flat rows: what this script measures and labels (Field() alone is tiled since round 4)
n = 1 << 24
def rate(a, b, c, reps=40):
    for _ in range(5): F.modmul(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
sets = []
for i in range(4):
    a = F.uniform(n, array=0); b = F.uniform(n, array=1); c = torch.empty_like(a)
    sets.append(("separate %d" % i, a, b, c))
    for skew in ((0, 0), (512, 1024), (8192, 16384)):
        slab = torch.empty(3 * 5 * n + 2 * (skew[0] + skew[1]) + 64, dtype=torch.int64, device="cuda")
        sa = slab[:5 * n].view(5, n)
        sb = slab[5 * n + skew[0]: 10 * n + skew[0]].view(5, n)
        sc = slab[10 * n + skew[0] + skew[1]: 15 * n + skew[0] + skew[1]].view(5, n)
        sa.copy_(a); sb.copy_(b)
        sets.append(("slab %d skew %s" % (i, skew), sa, sb, sc))
for rnd in range(2):
    for name, a, b, c in (sets if rnd == 0 else reversed(sets)):
        print("round %d  %-28s %.0f GB/s" % (rnd, name, rate(a, b, c)), flush=True)
