#!/usr/bin/env python3
"""throughput of the VALU-bound batched field kernels (modinv, modsqrt, modpro) per prime (GPU box), best of 3 at 2^22
elements; run once per product policy (default per-wave choice, MA_FORCE_EXACT=1) in child processes"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from modarith_amd.field import Field
    from modarith_amd.params import derive
    n = 1 << 22
    for name in sys.argv[2:] or ["X25519", "NIST256", "X448"]:
        fp = derive(name)
        F = Field(name, tile=None)          # flat rows, as the [N, n] indexing below assumes
        a = F.nres(F.uniform(n))
        for op in ("modinv", "modsqrt", "modpro"):
            fn = getattr(F, op)
            fn(a[:, :4096].contiguous()); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); fn(a); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            print("%-8s %-9s %-8s %7.2f ms  %.3e per s" % (os.environ.get("MA_TAG", ""), name, op, best * 1e3, n / best), flush=True)
else:
    for tag, extra in (("default", {}), ("exact", {"MA_FORCE_EXACT": "1"})):
        env = {k: v for k, v in os.environ.items() if k not in ("MA_FORCE_EXACT", "MA_FORCE_FAST")}
        env.update(extra, MA_TAG=tag)
        subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=env, check=True)
