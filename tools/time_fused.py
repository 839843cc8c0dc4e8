#!/usr/bin/env python3
"""fused ecn mul_get against the two-call form (ecn mul + ecn get) on the GPU box: rate of both, best of 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.edwards import Curve
only = [a.upper() for a in sys.argv[1:]]
for name, n in (("ED25519", 1 << 21), ("ED448", 1 << 19), ("NIST256", 1 << 20), ("SECP256K1", 1 << 20)):
    if only and name not in only:
        continue
    Ed = Curve(name)
    g = torch.Generator(device="cuda").manual_seed(3)
    e = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    k = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    P = Ed.mul(k, Ed.gen(n))
    Ed.mul_get(e[:4096].contiguous(), P[:, :, :4096].contiguous()); torch.cuda.synchronize()
    bf = bt = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); x, y, s = Ed.mul_get(e, P); torch.cuda.synchronize(); bf = min(bf, time.perf_counter() - t0)
    for _ in range(3):
        Q = P.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); Ed.mul(e, Q); wx, wy, _ = Ed.get(Q); torch.cuda.synchronize(); bt = min(bt, time.perf_counter() - t0)
    print("%s 2^%d: fused mul_get %.3e/s (%.1f ms)   mul + get %.3e/s (%.1f ms)   ratio %.2f   equal: %s" % (
        name, n.bit_length() - 1, n / bf, bf * 1e3, n / bt, bt * 1e3, bt / bf, bool(torch.equal(x, wx) and torch.equal(y, wy))), flush=True)

# fused double multiplication (verification pattern) against mul2 + get
for name2, n in (("ED25519", 1 << 20), ("ED448", 1 << 18), ("NIST256", 1 << 19), ("SECP256K1", 1 << 19)):
  if only and name2 not in only:
      continue
  Ed = Curve(name2)
  g = torch.Generator(device="cuda").manual_seed(4)
  rnd = lambda: torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=g)
  e, f = rnd(), rnd()
  P, Q = Ed.mul(rnd(), Ed.gen(n)), Ed.mul(rnd(), Ed.gen(n))
  Ed.mul2_get(e[:4096].contiguous(), P[:, :, :4096].contiguous(), f[:4096].contiguous(), Q[:, :, :4096].contiguous()); torch.cuda.synchronize()
  bf = bt = 1e9
  for _ in range(3):
      t0 = time.perf_counter(); x, y, s = Ed.mul2_get(e, P, f, Q); torch.cuda.synchronize(); bf = min(bf, time.perf_counter() - t0)
  for _ in range(3):
      t0 = time.perf_counter(); wx, wy, _ = Ed.get(Ed.mul2(e, P, f, Q)); torch.cuda.synchronize(); bt = min(bt, time.perf_counter() - t0)
  print(name2 + " 2^%d: fused mul2_get" % (n.bit_length() - 1) + " %.3e/s (%.1f ms)   mul2 + get %.3e/s (%.1f ms)   ratio %.2f   equal: %s" % (
      n / bf, bf * 1e3, n / bt, bt * 1e3, bt / bf, bool(torch.equal(x, wx) and torch.equal(y, wy))), flush=True)

# fused generator multiplication (key generation / signing pattern) against gen + mul + get and against mul_get on the generator
for name3, n in (("NIST256", 1 << 21), ("SECP256K1", 1 << 21), ("ED25519", 1 << 22), ("ED448", 1 << 20)):
    if only and name3 not in only:
        continue
    Ed = Curve(name3)
    g = torch.Generator(device="cuda").manual_seed(5)
    e = torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    Ed.mulgen_get(e[:4096].contiguous()); torch.cuda.synchronize()
    bf = bt = bm = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); x, y, s = Ed.mulgen_get(e); torch.cuda.synchronize(); bf = min(bf, time.perf_counter() - t0)
    for _ in range(2):
        t0 = time.perf_counter(); wx, wy, _ = Ed.get(Ed.mul(e, Ed.gen(n))); torch.cuda.synchronize(); bt = min(bt, time.perf_counter() - t0)
        t0 = time.perf_counter(); mx, my, _ = Ed.mul_get(e, Ed.gen(n)); torch.cuda.synchronize(); bm = min(bm, time.perf_counter() - t0)
    print("%s 2^%d: fused mulgen_get %.3e/s (%.1f ms)   gen + mul + get %.3e/s (%.1f ms)   ratio %.2f   (gen + mul_get %.3e/s)   equal: %s" % (
        name3, n.bit_length() - 1, n / bf, bf * 1e3, n / bt, bt * 1e3, bt / bf, n / bm, bool(torch.equal(x, wx) and torch.equal(y, wy))), flush=True)

# e*G + f*Q (verification) against the general fused mul2_get with P = G and against gen + mul2 + get
for name4, n in (("NIST256", 1 << 20), ("SECP256K1", 1 << 20), ("ED25519", 1 << 21), ("ED448", 1 << 19)):
    if only and name4 not in only:
        continue
    Ed = Curve(name4)
    g = torch.Generator(device="cuda").manual_seed(6)
    rnd = lambda: torch.randint(0, 256, (n, Ed.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    e, f = rnd(), rnd()
    Q = Ed.mul(rnd(), Ed.gen(n))
    Ed.mulgen2_get(e[:4096].contiguous(), f[:4096].contiguous(), Q[:, :, :4096].contiguous()); torch.cuda.synchronize()
    bf = bt = bm = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); x, y, s = Ed.mulgen2_get(e, f, Q); torch.cuda.synchronize(); bf = min(bf, time.perf_counter() - t0)
    for _ in range(2):
        t0 = time.perf_counter(); mx, my, _ = Ed.mul2_get(e, Ed.gen(n), f, Q); torch.cuda.synchronize(); bm = min(bm, time.perf_counter() - t0)
    m = n // 4
    t0 = time.perf_counter(); wx, wy, _ = Ed.get(Ed.mul2(e[:m].contiguous(), Ed.gen(m), f[:m].contiguous(), Q[:, :, :m].contiguous())); torch.cuda.synchronize(); bt = time.perf_counter() - t0
    print("%s 2^%d: fused mulgen2_get %.3e/s (%.1f ms)   gen + mul2_get %.3e/s (ratio %.2f)   gen + mul2 + get %.3e/s (ratio %.2f)   equal: %s" % (
        name4, n.bit_length() - 1, n / bf, bf * 1e3, n / bm, bm / bf, m / bt, (bt / m) / (bf / n), bool(torch.equal(x, mx) and torch.equal(y, my) and torch.equal(x[:m], wx))), flush=True)
