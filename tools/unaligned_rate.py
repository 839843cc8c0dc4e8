import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from modarith_amd.field import Field
for P in ("X25519", "NIST256", "X448"):
    F = Field(P); n = (1 << 24) + 2
    a = F.uniform(n); b = F.uniform(n, array=1); c = torch.empty_like(a)
    def rate(fn, reps=10):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    m = 1 << 24
    al = rate(lambda: F.modmul(a[:, :m], b[:, :m], out=c[:, :m]))
    un = rate(lambda: F.modmul(a[:, 1:m + 1], b[:, 1:m + 1], out=c[:, 1:m + 1]))
    nb = 3 * 8 * F.N * m
    print("%s modmul aligned %.0f GB/s   unaligned (8 B per lane) %.0f GB/s" % (P, nb / al / 1e6, nb / un / 1e6), flush=True)
