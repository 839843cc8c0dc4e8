"""Random soak of the round-5 Weierstrass fused forms (GPU box): for a number of seeds, 2^18 random scalars and random projective points
per curve through mul_get, mulgen_get, mulgen2_get and mul2_get against the bit-exact call-by-call forms (ecn mul / mul2 + ecn get, the
reference's complete formulas, pinned to the oracle by the GPU suite).  Scalars are drawn from three populations: uniform 256-bit,
short (64-136 bits: leading zero windows, the sizes of an endomorphism split), and within 2^20 of the group order / of 2^256.
    python tools/fuzz_weier.py [seeds] [log2_n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from modarith_amd.edwards import Curve  # noqa: E402

ORDER = {"NIST256": 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551,
         "SECP256K1": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141}


def scalars(n, q, gen):
    e = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    kind = torch.randint(0, 8, (n,), device="cuda", generator=gen)
    keep = torch.randint(8, 18, (n,), device="cuda", generator=gen)                  # bytes kept by the short population
    col = torch.arange(32, device="cuda").unsqueeze(0)
    short = (kind == 1).unsqueeze(1) & (col < (32 - keep).unsqueeze(1))
    e = torch.where(short, torch.zeros_like(e), e)
    # near q / near 2^256: q - d, q + d, 2^256 - d for small d -- built on the host for the few records of that population
    idx = torch.nonzero(kind == 2).flatten().cpu().tolist()
    rows = []
    g = torch.Generator().manual_seed(int(gen.initial_seed()) + 1)
    ds = torch.randint(0, 1 << 20, (len(idx),), generator=g).tolist()
    for k, d in enumerate(ds):
        v = (q - d, q + d, (1 << 256) - 1 - d)[k % 3]
        rows.append(list(v.to_bytes(32, "big")))
    if idx:
        e[torch.tensor(idx, device="cuda")] = torch.tensor(rows, dtype=torch.uint8, device="cuda")
    return e.contiguous()


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 18
    n = 1 << lg
    bad = 0
    for name in ("NIST256", "SECP256K1"):
        W = Curve(name)
        q = ORDER[name]
        for seed in range(seeds):
            gen = torch.Generator(device="cuda").manual_seed(1000 + seed)
            e, f, k = scalars(n, q, gen), scalars(n, q, gen), scalars(n, q, gen)
            P = W.mul(k, W.gen(n))
            P[:, :, ::4099] = W.inf((n + 4098) // 4099)
            checks = []
            x, y, _ = W.mul_get(e, P)
            wx, wy, _ = W.get(W.mul(e, P.clone()))
            checks.append(("mul_get", torch.equal(x, wx) and torch.equal(y, wy)))
            x, y, _ = W.mulgen_get(e)
            wx, wy, _ = W.get(W.mul(e, W.gen(n)))
            checks.append(("mulgen_get", torch.equal(x, wx) and torch.equal(y, wy)))
            m = n // 2
            em, fm, Pm = e[:m].contiguous(), f[:m].contiguous(), P[:, :, :m].contiguous()
            x, y, _ = W.mulgen2_get(em, fm, Pm)
            wx, wy, _ = W.get(W.mul2(em, W.gen(m), fm, Pm))
            checks.append(("mulgen2_get", torch.equal(x, wx) and torch.equal(y, wy)))
            Qm = P[:, :, m:].contiguous()
            x, y, _ = W.mul2_get(em, Pm, fm, Qm)
            wx, wy, _ = W.get(W.mul2(em, Pm, fm, Qm))
            checks.append(("mul2_get", torch.equal(x, wx) and torch.equal(y, wy)))
            bad += sum(not ok for _, ok in checks)
            print("%-10s seed %2d  n = 2^%d  %s" % (name, seed, lg, "  ".join("%s %s" % (c, "ok" if ok else "DIFFERS") for c, ok in checks)), flush=True)
    print("records per form and curve: %d; forms that differed: %d" % (seeds * n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
