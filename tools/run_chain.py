#!/usr/bin/env python3
"""the fused four-call chain of bench.py (modsqr(modmul(modadd(a,b), modsub(a,b))), modarith_amd/fuse.py bench_chain) and the headline
modmul on the same tiled 2^24-element operands, a few launches each: the program profiled by tools/gpu_r04_pmc.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
from modarith_amd.fuse import bench_chain
n = 1 << 24
F = Field("X25519", tile=4096)
a, b = F.uniform(n, seed=42, array=0), F.uniform(n, seed=42, array=1)
c, z = torch.empty_like(a), torch.empty_like(a)
fz = bench_chain("X25519").build()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    F.modmul(a, b, out=c)
    fz(a, b, out=[z])
    F.modadd(a, b, out=c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("modmul", lambda: F.modmul(a, b, out=c)), ("chain", lambda: fz(a, b, out=[z])), ("modadd", lambda: F.modadd(a, b, out=c))):
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s %.4f ms  %.1f GB/s of 120 B per element" % (name, ms, 120 * n / ms / 1e6))
