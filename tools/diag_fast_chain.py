#!/usr/bin/env python3
"""time_protocol chains (fused in registers) for one prime: run plain and with MA_FORCE_FAST=1 (child processes) and
count the lanes whose results differ"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    from modarith_amd.field import Field
    from modarith_amd.params import derive
    P = sys.argv[1]
    fp = derive(P); F = Field(P)
    torch.manual_seed(1)
    n = 4096
    x = torch.randint(0, 1 << fp.radix, (fp.nlimbs, n), dtype=torch.int64, device="cuda")
    x[fp.nlimbs - 1] &= (1 << (fp.n - fp.radix * (fp.nlimbs - 1) - 1)) - 1
    y = x.flip(1).contiguous()
    out = {}
    for kind in ("modmul", "modsqr", "modinv"):
        out[kind] = F.time_protocol(kind, x, y if kind == "modmul" else None, 1).cpu()
    torch.save(out, sys.argv[3])
else:
    P = sys.argv[1]
    import torch
    res = []
    for fast in ("0", "1"):
        f = "/tmp/diag_%s_%s.pt" % (P, fast)
        subprocess.run([sys.executable, __file__, P, "child", f], env=dict(os.environ, MA_FORCE_FAST=fast), check=True)
        res.append(torch.load(f))
    for kind in res[0]:
        d = (res[0][kind] != res[1][kind]).any(dim=0)
        print(P, kind, "lanes differing: %d of %d" % (int(d.sum()), d.numel()), "first:", d.nonzero()[:5].flatten().tolist())
