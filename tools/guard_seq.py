import sys, os
sys.path.insert(0, "/root/repo")
import torch
from modarith_amd.edwards import Curve
g = torch.Generator(device="cuda").manual_seed(1)
C = Curve("NIST256"); n = 1 << 19
e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
G = C.gen(n)
Pr = C.mul(torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g), C.gen(n))
for label, src in (("generator in every lane", G), ("random points", Pr), ("generator in every lane", G)):
    ts = []
    for i in range(8):
        P = src.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); C.mul(e, P); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("library ecn mul NIST256, %s: " % label + " ".join("%.2f" % t for t in ts), "ms", flush=True)
