import sys, time, torch
sys.path.insert(0, '.')
from modarith_amd.edwards import Curve
for name in ("ED25519", "NIST256", "ED448", "SECP256K1"):
    C = Curve(name)
    n = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(1)
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    P = C.mul(e, C.gen(n))
    C.get(P); torch.cuda.synchronize()
    t0 = time.perf_counter(); C.get(P); torch.cuda.synchronize(); t = time.perf_counter() - t0
    t0 = time.perf_counter(); Q = C.affine(P.clone()); torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print("%-10s get %.3e points/s   affine %.3e/s" % (name, n / t, n / t2))
