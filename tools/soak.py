#!/usr/bin/env python3
"""Soak runs on the GPU box (not part of the pytest suites: minutes, not seconds).

  soak.py field   exact vs split/chain products on 2^22 elements per prime whose limbs are drawn from the contract's
                  edge classes (0, 1, 2^R-1, 2^R, 2^(R+1)-1, 2^(R+2)-1, random), plus 48-operation chains from
                  in-range values: the product policies must agree bit for bit (checksums compared across three child processes: MA_FORCE_EXACT=1, MA_FORCE_FAST=1, default)
  soak.py curves  2^13 .. 2^16 random scalars x random points for each of the eleven curves, fused ecn mul on the GPU against the CPU oracle,
                  projective limbs compared
  soak.py mul2    2^10 .. 2^12 (scalar pair, point pair) records per curve through Curve.mul2(exact=True) against the oracle's mul2, projective limbs
  soak.py fused   2^18 .. 2^20 random (scalar, projective point) pairs per fused curve (ED25519, ED448, NIST256, SECP256K1): mul_get against mul + get and
                  mul2_get against mul2 + get (the two-call forms are the ones `soak.py curves` pins to the oracle), the generator forms mulgen_get / mulgen2_get against
                  those, rfc7748 on the base point against the ladder; bytes compared
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def field_child(out):
    import json, torch
    from modarith_amd.field import Field
    from modarith_amd.params import derive
    from modarith_amd import emit
    res = {}
    n = 1 << 22
    for name in emit.BUILT_PRIMES:
        fp = derive(name)
        if emit.split_point(fp) == 0:
            continue
        F = Field(name, tile=None)          # flat rows, as the [N, n] indexing below assumes
        R = fp.radix
        g = torch.Generator(device="cuda").manual_seed(1234)
        edges = torch.tensor([0, 1, (1 << R) - 1, 1 << R, (1 << (R + 1)) - 1, (1 << (R + 2)) - 1, 0x5555555555555 & ((1 << R) - 1)], dtype=torch.int64, device="cuda")
        def draw():
            cls = torch.randint(0, 10, (fp.nlimbs, n), device="cuda", generator=g)
            rnd = torch.randint(0, 1 << R, (fp.nlimbs, n), dtype=torch.int64, device="cuda", generator=g)
            return torch.where(cls < 7, edges[cls.clamp(max=6)], rnd)
        a, b = draw(), draw()
        outs = [F.modmul(a, b), F.modsqr(a), F.modmul(b, b)]
        if fp.montgomery:
            outs += [F.nres(a), F.redc(a)]
        # chained, from in-range values (tight limbs, value below 2^Nbits): results fed back as the functions leave them.
        # (Edge-class limbs are integers up to 4*2^(N*Radix), far outside the < 2p domain of the reference's functions;
        # their single products agree above, but their outputs are not valid inputs any more.)
        x = torch.randint(0, 1 << R, (fp.nlimbs, n), dtype=torch.int64, device="cuda", generator=g)
        x[fp.nlimbs - 1] &= (1 << max(fp.n - R * (fp.nlimbs - 1) - 1, 1)) - 1
        y = x.flip(1).contiguous()
        c = F.modmul(x, y)
        for _ in range(16):
            c = F.modmul(c, y); c = F.modsqr(c); y = F.modadd(c, x)
        outs.append(c)
        res[name] = [[int(o.sum().item()) & (2**64 - 1), int(o.flatten().cumsum(0)[-1].item()) & (2**64 - 1),
                      int((o ^ (o >> 17)).sum().item()) & (2**64 - 1)] for o in outs]
    json.dump(res, open(out, "w"))


def field():
    """three child processes, one per product policy (the switches are process-static): MA_FORCE_EXACT=1 (128-bit
    products: the reference arithmetic), MA_FORCE_FAST=1 (split / chain products, unguarded) and the default per-wave
    vote.  All edge classes are inside the split contract, so the default leg takes the split path too; the exact
    leg is the one that differs in code.  (Against the CPU oracle: tests/test_gpu_round2.py, three primes.)"""
    import json
    legs = (("exact", {"MA_FORCE_EXACT": "1"}), ("fast", {"MA_FORCE_FAST": "1"}), ("default", {}))
    outs = []
    for tag, extra in legs:
        f = "/tmp/soak_field_%s.json" % tag
        env = {k: v for k, v in os.environ.items() if k not in ("MA_FORCE_EXACT", "MA_FORCE_FAST")}
        env.update(extra)
        subprocess.run([sys.executable, __file__, "field-child", f], env=env, check=True)
        outs.append(json.load(open(f)))
    names = ["modmul(a,b)", "modsqr(a)", "modmul(b,b)", "nres(a)|chain", "redc(a)", "chain"]
    bad = []
    for k in outs[0]:
        for leg, o in zip(legs[1:], outs[1:]):
            if outs[0][k] != o[k]:
                bad.append((k, leg[0]))
                print(k, leg[0], "differs from exact in:", [names[i] if len(o[k]) == 6 or i < 3 else "chain" for i in range(len(o[k])) if outs[0][k][i] != o[k][i]])
    print("field soak: %d primes, 2^22 edge-class elements each, exact == split/chain == default: %s" % (len(outs[0]), "ALL EQUAL" if not bad else "MISMATCH " + str(bad)))
    return 1 if bad else 0


def curves():
    import ctypes, numpy as np, torch
    from modarith_amd.edwards import Curve
    from tests.oracle_binding import load_oracle
    from tests.util import vp
    o = load_oracle(build=not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")))
    rc = 0
    for name, lg in (("ED25519", 16), ("NIST256", 16), ("ED448", 16), ("NIST384", 16), ("SECP256K1", 15), ("NUMS256W", 14), ("NUMS256E", 14),
                     ("ED248", 14), ("ED376", 13), ("NIST521", 13), ("ED500", 13)):
        n = 1 << lg
        C = Curve(name)
        g = torch.Generator(device="cuda").manual_seed(77)
        e0 = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
        P = C.mul(e0, C.gen(n))                       # random points (GPU), then the run under test
        e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
        e[:16] = 0; e[16:32] = 255                    # corner scalars
        hp = np.ascontiguousarray(P.cpu().numpy().view(np.uint64)).reshape(3 * C.N, n)
        he = np.ascontiguousarray(e.cpu().numpy())
        want = hp.copy()
        o.lib.__getattr__("ecn_%s_batch_mul" % name.lower())(vp(he), vp(want), n, n)
        got = C.mul(e, P.clone()).cpu().numpy().view(np.uint64).reshape(3 * C.N, n)
        ok = bool(np.array_equal(got, want))
        print("curve soak %-9s 2^%d scalar multiplications, projective limbs vs oracle: %s" % (name, lg, "EQUAL" if ok else "MISMATCH"), flush=True)
        rc |= 0 if ok else 1
    return rc


def mul2_exact():
    """Curve.mul2(exact=True) -- the reference's joint-sparse-form walk, lanes diverging -- against the oracle's mul2 (which
    reproduces the reference's limbs, tests/test_curveref_oracle.py): 2^12 (2^10 for the widest fields) pairs per curve, projective limbs"""
    import ctypes, numpy as np, torch
    from modarith_amd.edwards import Curve
    from tests.oracle_binding import load_oracle
    o = load_oracle(build=not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")))
    rc = 0
    for name, lg in (("ED25519", 12), ("NIST256", 12), ("SECP256K1", 12), ("NUMS256W", 12), ("NUMS256E", 12), ("ED248", 12), ("ED448", 11), ("NIST384", 11),
                     ("ED376", 11), ("NIST521", 10), ("ED500", 10)):
        n = 1 << lg
        C = Curve(name)
        c = name.lower()
        Pt, nb = o.ed[c]
        g = torch.Generator(device="cuda").manual_seed(91)
        rnd = lambda: torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
        P, Q = C.mul(rnd(), C.gen(n)), C.dbl(C.mul(rnd(), C.gen(n)))
        e, f = rnd(), rnd()
        e[:64, :-2] = 0                               # short scalars: their walks start hundreds of digits later than their neighbours'
        f[32:96, :-1] = 0
        R = C.mul2(e, P, f, Q, exact=True).cpu().numpy().view(np.uint64)
        hp, hq, he, hf = P.cpu().numpy().view(np.uint64), Q.cpu().numpy().view(np.uint64), e.cpu().numpy(), f.cpu().numpy()
        fn = o.ecn(c, "mul2")
        bad = 0
        for j in range(n):
            p, q, r = Pt(), Pt(), Pt()
            for k, nm in enumerate("xyz"):
                for i in range(C.N):
                    getattr(p, nm)[i] = int(hp[k, i, j]); getattr(q, nm)[i] = int(hq[k, i, j])
            fn(bytes(he[j]), ctypes.byref(p), bytes(hf[j]), ctypes.byref(q), ctypes.byref(r))
            bad += any([int(v) for v in R[k, :, j]] != list(getattr(r, nm)) for k, nm in enumerate("xyz"))
        print("mul2 soak %-9s 2^%d double multiplications (exact form), projective limbs vs oracle: %s" % (name, lg, "EQUAL" if not bad else "%d MISMATCHES" % bad), flush=True)
        rc |= 1 if bad else 0
    return rc


def fused():
    import torch
    from modarith_amd.edwards import Curve
    rc = 0
    for name, lg in (("ED25519", 20), ("ED448", 18), ("NIST256", 20), ("SECP256K1", 20)):
        n = 1 << lg
        C = Curve(name)
        g = torch.Generator(device="cuda").manual_seed(78)
        rnd = lambda m: torch.randint(0, 256, (m, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
        P = C.mul(rnd(n), C.gen(n))
        e = rnd(n)
        e[:16] = 0; e[16:32] = 255
        x, y, _ = C.mul_get(e, P)
        wx, wy, _ = C.get(C.mul(e, P.clone()))
        ok = bool(torch.equal(x, wx) and torch.equal(y, wy))
        m = n // 4
        Q = C.mul(rnd(m), C.gen(m))
        Pm, em, fm = P[:, :, :m].contiguous(), e[:m].contiguous(), rnd(m)
        x, y, _ = C.mul2_get(em, Pm, fm, Q)
        wx, wy, _ = C.get(C.mul2(em, Pm, fm, Q))
        ok2 = bool(torch.equal(x, wx) and torch.equal(y, wy))
        # generator forms against the general fused kernels on the generator (those are pinned to the two-call forms above)
        x, y, _ = C.mulgen_get(e)
        gx, gy, _ = C.mul_get(e, C.gen(n))
        ok3 = bool(torch.equal(x, gx) and torch.equal(y, gy))
        x, y, _ = C.mulgen2_get(em, fm, Q)
        gx, gy, _ = C.mul2_get(em, C.gen(m), fm, Q)
        ok4 = bool(torch.equal(x, gx) and torch.equal(y, gy))
        print("fused soak %-8s mul_get 2^%d: %s   mul2_get 2^%d: %s   mulgen_get 2^%d: %s   mulgen2_get 2^%d: %s" % (
            name, lg, "EQUAL" if ok else "MISMATCH", lg - 2, "EQUAL" if ok2 else "MISMATCH", lg, "EQUAL" if ok3 else "MISMATCH",
            lg - 2, "EQUAL" if ok4 else "MISMATCH"), flush=True)
        rc |= 0 if (ok and ok2 and ok3 and ok4) else 1
    from modarith_amd.field import rfc7748, rfc7748_base
    for curve, nb, lg in (("X25519", 32, 21), ("X448", 56, 19)):
        n = 1 << lg
        g = torch.Generator(device="cuda").manual_seed(79)
        k = torch.randint(0, 256, (n, nb), dtype=torch.uint8, device="cuda", generator=g)
        u = torch.zeros((n, nb), dtype=torch.uint8, device="cuda"); u[:, 0] = 9 if curve == "X25519" else 5
        ok = bool(torch.equal(rfc7748_base(curve, k), rfc7748(curve, k, u)))
        print("fused soak %-8s rfc7748 on the base point 2^%d, fixed-base kernel vs ladder: %s" % (curve, lg, "EQUAL" if ok else "MISMATCH"), flush=True)
        rc |= 0 if ok else 1
    return rc


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "field"
    if mode == "field-child":
        field_child(sys.argv[2])
    elif mode == "field":
        sys.exit(field())
    elif mode == "mul2":
        sys.exit(mul2_exact())
    elif mode == "fused":
        sys.exit(fused())
    else:
        sys.exit(curves())
