#!/usr/bin/env python3
"""Is the fused chain's in-bench reading (0.69-0.78 of the HBM peak) against 0.815 in a fresh process a clock effect?  The chain is
co-limited by VALU issue (docs/fused_chains.md), the headline modmul is not: time both (20 launches each) fresh, after N seconds of
sustained headline launches, after a pause, and on a second output buffer.  Prints one line per phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
from modarith_amd.fuse import bench_chain
n = 1 << 24
F = Field("X25519", tile=4096)
a, b = F.uniform(n, seed=42, array=0), F.uniform(n, seed=42, array=1)
c, z = torch.empty_like(a), torch.empty_like(a)
fz = bench_chain("X25519").build()
fz(a, b, out=[z]); F.modmul(a, b, out=c); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, k=20):
    e0.record()
    for _ in range(k):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
def phase(name):
    m = t(lambda: F.modmul(a, b, out=c)); ch = t(lambda: fz(a, b, out=[z])); ch2 = t(lambda: fz(a, b, out=[c]))
    print("%-34s modmul %.4f ms (%.3f)   chain->z %.4f ms (%.3f)   chain->c %.4f ms (%.3f)" % (name, m, 120 * n / m / 8e9, ch, 120 * n / ch / 8e9, ch2, 120 * n / ch2 / 8e9), flush=True)
phase("fresh")
for secs in (1, 3):
    t0 = time.time()
    while time.time() - t0 < secs:
        t(lambda: F.modmul(a, b, out=c), 200)
    phase("after %d s of modmul launches" % secs)
t0 = time.time()
while time.time() - t0 < 3:
    t(lambda: fz(a, b, out=[z]), 200)
phase("after 3 s of chain launches")
time.sleep(3)
phase("after 3 s idle")
for _ in range(3):
    phase("again")
