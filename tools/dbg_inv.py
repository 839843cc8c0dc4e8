import sys, torch
sys.path.insert(0, '.')
from modarith_amd.field import Field
F = Field("X25519")
n = 3 * 16384 + 1237
x = F.nres(F.uniform(n, seed=11, array=3))
want = torch.empty_like(x)
for lo in range(0, n, 8192):
    hi = min(n, lo + 8192)
    want[:, lo:hi] = F.modinv(x[:, lo:hi].contiguous())
got = F.modinv(x)
bad = (got != want).any(dim=0).nonzero().flatten()
print("plain uniform: differing elements", bad.numel(), bad[:10].tolist())
fp = F.params
zf = [fp.to_limbs(0), fp.to_limbs(fp.p), fp.to_limbs(2 * fp.p)]
for k, name in enumerate(("0", "p", "2p")):
    y = x.clone()
    y[:, 100] = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in zf[k]], dtype=torch.int64)
    w = torch.empty_like(y)
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        w[:, lo:hi] = F.modinv(y[:, lo:hi].contiguous())
    g = F.modinv(y)
    bad = (g != w).any(dim=0).nonzero().flatten()
    print("zero form", name, "differing", bad.numel(), bad[:10].tolist(), "is0", int(F.modis0(y)[100]), "per-elem out", F.to_limbs(w[:, 100:101].contiguous()), "simul out", F.to_limbs(g[:, 100:101].contiguous()))
y = x.clone(); y[0, 5000] = -1
w = torch.empty_like(y)
for lo in range(0, n, 8192):
    hi = min(n, lo + 8192)
    w[:, lo:hi] = F.modinv(y[:, lo:hi].contiguous())
g = F.modinv(y)
bad = (g != w).any(dim=0).nonzero().flatten()
print("ooc limb: differing", bad.numel(), bad[:10].tolist())
