#!/usr/bin/env python3
"""one prime, one process, one set of buffers: the multiplying kernels against their streaming controls on the same arrays
(modmul vs modadd: two reads, one write; modsqr / modmuls vs modcpy: one read, one write), interleaved rounds, median.
   python tools/stream_controls.py X448 [pad_elements] [log2n] [tile]     (tile > 0: the tiled layout, pad ignored)
env knobs are the library's (MA_FORCE_FAST, MA_FORCE_EXACT, MA_MAX_BLOCKS); pad > 0 takes the batch as a view of a wider
allocation, i.e. a limb stride of n + pad elements."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
name = sys.argv[1] if len(sys.argv) > 1 else "X448"
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 24)
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
F = Field(name, tile=tile or None)
def alloc(src=None):
    if tile:
        return src.clone() if src is not None else F.empty(n)
    t = torch.empty((F.N, n + pad), dtype=torch.int64, device="cuda")[:, :n]
    if src is not None:
        t.copy_(src)
    return t
a = alloc(F.uniform(n, seed=42, array=0)); b = alloc(F.uniform(n, seed=42, array=1)); c = alloc()
b0 = [int(v) for v in F.to_limbs(F.uniform(1, seed=42, array=1))[0]]
ops = [("modadd", 3, lambda: F.modadd(a, b, out=c)), ("modmul", 3, lambda: F.modmul(a, b, out=c)),
       ("modcpy", 2, lambda: F.modcpy(a, out=c)), ("modsqr", 2, lambda: F.modsqr(a, out=c)), ("modmuls", 2, lambda: F.modmuls(a, b0, out=c))]
res = {k: [] for k, _, _ in ops}
for rnd in range(7):
    for k, nb, fn in ops:
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(nb * 8 * F.N * n / (e0.elapsed_time(e1) / 10) / 1e6)
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("MA_FORCE_FAST", "MA_FORCE_EXACT", "MA_MAX_BLOCKS") if k in os.environ)
print("%-8s %s n 2^%d %-28s " % (name, ("tile %-5d" % tile) if tile else ("pad %-6d" % pad), n.bit_length() - 1, tag) + "  ".join("%s %4.0f" % (k, statistics.median(v)) for k, v in res.items()), flush=True)
