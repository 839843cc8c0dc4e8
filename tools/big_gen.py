import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from modarith_amd.edwards import Curve
from modarith_amd.field import rfc7748, rfc7748_base
n = 1 << 24
for name in ("ED25519", "NIST256"):
    C = Curve(name)
    g = torch.Generator(device="cuda").manual_seed(11)
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    t0 = time.perf_counter(); x, y, _ = C.mulgen_get(e); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = True
    for lo in range(0, n, n // 4):          # four slices of 2^20 against the general fused kernel
        sl = slice(lo + 12345, lo + 12345 + (1 << 20))
        gx, gy, _ = C.mul_get(e[sl].contiguous(), C.gen(1 << 20))
        ok &= bool(torch.equal(x[sl], gx) and torch.equal(y[sl], gy))
    gx, gy, _ = C.mul_get(e[-4099:].contiguous(), C.gen(4099))
    ok &= bool(torch.equal(x[-4099:], gx) and torch.equal(y[-4099:], gy))
    print("%s mulgen_get 2^24 scalars: %.3e/s, slices vs mul_get on G: %s" % (name, n / dt, "EQUAL" if ok else "MISMATCH"), flush=True)
k = torch.randint(0, 256, (1 << 25, 32), dtype=torch.uint8, device="cuda")
t0 = time.perf_counter(); pk = rfc7748_base("X25519", k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
u = torch.zeros((1 << 20, 32), dtype=torch.uint8, device="cuda"); u[:, 0] = 9
ok = bool(torch.equal(pk[-(1 << 20):], rfc7748("X25519", k[-(1 << 20):].contiguous(), u)) and torch.equal(pk[:1 << 20], rfc7748("X25519", k[:1 << 20].contiguous(), u)))
print("X25519 public keys 2^25: %.3e/s, first / last 2^20 vs ladder: %s" % ((1 << 25) / dt, "EQUAL" if ok else "MISMATCH"))
