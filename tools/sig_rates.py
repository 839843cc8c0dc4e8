"""End-to-end rates of the reference's signature programs over batches (GPU box): NIST256_SIGN / NIST256_VERIFY (nist256.c:196-260) with
every gel / point operation on the device through the batched API -- the steps of examples/batch_signatures.py on device-resident
tensors (hashes are inputs: host byte work outside the arithmetic path) -- timed as a whole and with the fused curve kernel alone, so
that the share of the group-order arithmetic around it (modimp, modinv, modmul, modexp, ecn set) shows.
    python tools/sig_rates.py [log2_n]     ->  one line per flow, JSON at the end (gpurun_out/sig_rates.json)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from modarith_amd.edwards import Curve  # noqa: E402
from modarith_amd.field import Field  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    n = 1 << lg
    out = {"n": n}
    for name in ("NIST256", "SECP256K1"):
        C = Curve(name)
        if name == "NIST256":
            G = Field("NIST256Q", tile=None)
        else:
            # the build has no group-order field of secp256k1: the generator mode makes one (a plug-in compiled in seconds -- what the reference's
            # `python3 monty.py 64 <order>` is for a new modulus, curve.py:324-329); without hipcc at hand: the curve part only
            try:
                from modarith_amd import generate as _gen
                _gen.generate("SECP256K1Q=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", family="monty")
                G = Field("SECP256K1Q", tile=None)
            except Exception as ex:                                             # noqa: BLE001
                print("secp256k1 group-order field not generated (%s): curve part only" % ex)
                G = None
        gen = torch.Generator(device="cuda").manual_seed(7)
        rnd = lambda: torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
        prv, thm, ran = rnd(), rnd(), rnd()
        if G is None:
            # no group-order field of secp256k1 in the build: the curve part only
            u, v = rnd(), rnd()
            Q = C.mul(prv, C.gen(n))
            t = timed(lambda: C.mulgen2_get(u, v, Q))
            out[name] = {"verify_curve_part_per_s": n / t}
            t = timed(lambda: C.mulgen_get(ran, want_y=False))
            out[name]["sign_curve_part_per_s"] = n / t
            print("%-10s curve part of a verification %.3e/s, of a signature %.3e/s" % (name, out[name]["verify_curve_part_per_s"], out[name]["sign_curve_part_per_s"]))
            continue
        # key pairs and signatures to verify (NIST256_KEY_PAIR, NIST256_SIGN)
        pubx, puby, _ = C.mulgen_get(prv)

        def sign():
            e, _ = G.modimp(thm)
            s, _ = G.modimp(prv)
            k, _ = G.modimp(ran)
            h = G.modexp(k)
            x, _, _ = C.mulgen_get(h, want_y=False)
            kinv = G.modinv(k)
            r, _ = G.modimp(x)
            G.modmul(s, r, s)
            G.modadd(s, e, s)
            G.modmul(s, kinv, s)
            return G.modexp(r), G.modexp(s)

        sr, ss = sign()

        def verify():
            e, _ = G.modimp(thm)
            r, r_ok = G.modimp(sr)
            s, s_ok = G.modimp(ss)
            ok = (r_ok != 0) & (s_ok != 0) & (G.modis0(r) == 0) & (G.modis0(s) == 0)
            sinv = G.modinv(s)
            v = G.modexp(G.modmul(r, sinv))
            u = G.modexp(G.modmul(sinv, e))
            Q = C.set(None, pubx, puby)
            x, y, _ = C.mulgen2_get(u, v, Q)
            inf = (x == 0).all(dim=1) & (y[:, :-1] == 0).all(dim=1) & (y[:, -1] == 1)
            e2, _ = G.modimp(x)
            return ok & ~inf & (G.modcmp(r, e2) != 0)

        ok = verify()
        assert bool(ok.all()), "signatures made here must verify"
        bad = ss.clone()
        bad[::2, 31] ^= 1
        keep = ss
        ss = bad
        okb = verify()
        ss = keep
        assert not bool(okb[::2].any()) and bool(okb[1::2].all()), "tampered signatures must not verify"
        tv, ts = timed(verify), timed(sign)
        Q = C.set(None, pubx, puby)
        u, v = rnd(), rnd()
        tk = timed(lambda: C.mulgen2_get(u, v, Q))
        tg = timed(lambda: C.mulgen_get(ran, want_y=False))
        out[name] = {"verify_per_s": n / tv, "sign_per_s": n / ts, "verify_curve_part_per_s": n / tk, "sign_curve_part_per_s": n / tg,
                     "verify_ms": tv * 1e3, "verify_curve_part_ms": tk * 1e3, "sign_ms": ts * 1e3, "sign_curve_part_ms": tg * 1e3}
        print("%-10s verify %.3e/s (%.2f ms, fused e G + f Q alone %.2f ms)   sign %.3e/s (%.2f ms, fused e G alone %.2f ms)"
              % (name, n / tv, tv * 1e3, tk * 1e3, n / ts, ts * 1e3, tg * 1e3))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/sig_rates.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
