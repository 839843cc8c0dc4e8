#!/usr/bin/env python3
"""does the streaming rate of the headline modmul depend on where its three arrays sit?  (GPU box)
Allocates several (a, b, c) triples in one process (earlier ones stay alive, so addresses differ), times each,
then re-times them in reverse order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
F = Field("X25519", tile=None)     # flat rows: what this script measures and labels (Field() alone is tiled since round 4)
n = 1 << 24
sets = []
def rate(a, b, c, reps=50):
    for _ in range(5): F.modmul(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): F.modmul(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    return 120.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
pads = [0, 0, 1 << 20, 3 << 20, 0, 7 << 19]
keep = []
for i, pad in enumerate(pads):
    if pad: keep.append(torch.empty(pad, dtype=torch.uint8, device="cuda"))
    a = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
    b = torch.randint(0, 1 << 51, (5, n), dtype=torch.int64, device="cuda")
    c = torch.empty_like(a)
    sets.append((a, b, c))
    print("set %d  a=%#x b=%#x c=%#x  %.0f GB/s" % (i, a.data_ptr(), b.data_ptr(), c.data_ptr(), rate(a, b, c)), flush=True)
for i in reversed(range(len(sets))):
    print("again set %d  %.0f GB/s" % (i, rate(*sets[i])), flush=True)
# one slab, three views at chosen offsets
slab = torch.empty(3 * 5 * n + (64 << 20) // 8, dtype=torch.int64, device="cuda")
for off_b, off_c in ((0, 0), (1 << 17, 2 << 17), (4096 // 8, 8192 // 8), (1 << 20, 2 << 20)):
    a = slab[:5 * n].view(5, n)
    b = slab[5 * n + off_b: 10 * n + off_b].view(5, n)
    c = slab[10 * n + off_b + off_c: 15 * n + off_b + off_c].view(5, n)
    a.copy_(sets[0][0]); b.copy_(sets[0][1])
    print("slab off_b=%d off_c=%d words  %.0f GB/s" % (off_b, off_c, rate(a, b, c)), flush=True)
