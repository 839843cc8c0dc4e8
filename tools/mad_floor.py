#!/usr/bin/env python3
"""Algorithmic multiply-add floors of the VALU-bound legs of bench.py  ->  profiles/mad_floor.json   (CPU, no inputs)

bench.py's VALU rooflines price a leg against the issue rate of v_mad_u64_u32 alone, with the count of multiply-adds the KERNEL
executes per record (mad_per_scalar, from its code object).  That count is the kernel's, not the algorithm's.  This table is the
algorithm's: the field multiplications (M), squarings (S) and small-constant multiplications (c) the leg's algorithm performs per
record, times the SCHOOLBOOK column products of the limb form the kernel computes on --

    n limbs of 32-bit words:   M = n^2     S = n (n + 1) / 2     c = n
    16 x 28-bit limbs (2^448 - 2^224 - 1: phi^2 = phi + 1 admits Karatsuba, csrc/fe28.h):   M = 192   S = 108   c = 16

-- and nothing else: no fold by 19 / 2^32 + 977, no Montgomery digit x prime-limb terms, no carries.  Those are real work, some of it
forced by the prime's shape, but a multiply-add spent on them is a multiply-add above the floor, and mad_per_scalar / floor says how
many there are (P-256's Montgomery form: 144 multiply-adds per product against the 100 column products).  The field-operation counts
follow the formulas the kernels implement (cited per leg; the bit-exact `ecn mul` / `mul2` legs are pinned to the reference's own
formulas, edwards.c:73-145, weierstrass.c:68-281, and its 4-bit fixed window, edwards.c:435-482).  Shared inversions count with
their share per record (one inversion per 32 records of a lane's column -- the share of a full chunk; smaller batches pay more, which only
moves the kernel's count further above the floor -- Montgomery's trick: 3M per record + 2M for two numerators).

A floor is a LOWER bound on the kernel's count: tests/test_bench_line.py asserts mad_per_scalar >= 0.98 floor for every leg whose
counter summary is in profiles/ (a floor above the measured count means a formula here is wrong, not that the kernel is magic).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Form:
    def __init__(self, name, M, S, c):
        self.name, self.M, self.S, self.c = name, M, S, c


F10 = Form("10 x 25.5/26-bit limbs (fe26.h, fh51.h, fm26.h, fk26.h, the half-limb products of field.h)", 100, 55, 10)
F16 = Form("16 x 28-bit limbs, Karatsuba on phi = 2^224 (fe28.h, fh56.h)", 192, 108, 16)


def ops(M=0, S=0, c=0):
    return {"M": float(M), "S": float(S), "c": float(c)}


def add(*terms):
    out = ops()
    for t in terms:
        for k in out:
            out[k] += t[k]
    return out


def times(k, t):
    return {a: k * v for a, v in t.items()}


# inversions x^(p-2) by the addition chains of the kernels (csrc/fe_finish.h, wn_export.h): squarings = bits - 1, a dozen products
INV = {"25519": ops(M=11, S=254), "448": ops(M=13, S=447), "p256": ops(M=13, S=255), "k256": ops(M=15, S=255)}


def shared_inv(which, per=32, numerators=2):
    """one record's share of a shared inversion (Montgomery's trick): 3M per record for the prefix products and the way back, one
    product per numerator, 1/per of the inversion"""
    return add(ops(M=3 + numerators), times(1.0 / per, INV[which]))


# ---- formulas ------------------------------------------------------------------------------------------------------------------------
LADDER_STEP = ops(M=5, S=4, c=1)                    # rfc7748.c:186-221: 5 modmul, 4 modsqr, 1 modmli per scalar bit
ED_DBL, ED_ADD = ops(M=3, S=4), ops(M=10, S=1)      # edwards.c:107-145 (dbl), 73-104 (add; the product by d is a modmul / one of the ten)
WS3_DBL, WS3_ADD = ops(M=10, S=3), ops(M=14)        # weierstrass.c a = -3, b a field constant: RCB alg. 6 (8M + 3S + 2 m_b), alg. 4 (12M + 2 m_b)
WS0_DBL, WS0_ADD = ops(M=6, S=2, c=1), ops(M=12, c=2)   # weierstrass.c a = 0, small b (secp256k1: 3b = 21 by modmli): RCB alg. 9 / 7
JAC_DBL = ops(M=3, S=5)                             # dbl-2001-b (wj26.h)
JAC_ADD, JAC_MADD = ops(M=11, S=5), ops(M=7, S=4)   # add-2007-bl, madd-2007-bl (wj26.h)
WS3_MADD_COMPLETE = ops(M=13)                       # RCB alg. 5, a = -3: 11M + 2 m_b
WS0_MADD_COMPLETE = ops(M=11, c=2)                  # RCB alg. 8, a = 0
ED_EXT_DBL, ED_EXT_DBL_T = ops(M=3, S=4), ops(M=4, S=4)     # dbl-2008-hwcd without / with T
ED_EXT_ADD_CACHED, ED_EXT_ADD_CACHED_NOT = ops(M=8), ops(M=7)   # add-2008-hwcd-3 on cached (Y+X, Y-X, 2dT, 2Z) with / without T out
ED_EXT_MADD = ops(M=7)                              # madd-2008-hwcd-3 on an affine cached entry (y+x, y-x, 2dxy)
ED_EXT_ADD = ops(M=9)                               # add-2008-hwcd-3, both points extended
# a = 1 (ED448: x^2 + y^2 = 1 + d x^2 y^2): the (Y - X)(Y' - X') trick of a = -1 is not available, A = X1 X2 and B = Y1 Y2 are separate
# products and M = (X1 + Y1)(X2 + Y2) a third (add-2008-hwcd with d T2 precomputed): one product more per addition
A1_MADD = ops(M=8)                                  # mixed, T out (csrc/ed28.h add_cached + add_tail)
A1_ADD_CACHED, A1_ADD_CACHED_NOT = ops(M=9), ops(M=8)
A1_ADD = ops(M=10)


def fixed_window(nbytes, dbl, addf, adds_per_window=1, tables=1):
    """the reference's signed 4-bit fixed window (edwards.c:435-482): table W[2..8] = 4 doublings + 3 additions; 2 Nbytes windows of four
    doublings and one addition.  mul2 of the batched API: two such multiplications that share their doublings (csrc/curve.h k_ed_mul2)"""
    return add(times(4 * tables + 8 * nbytes, dbl), times(3 * tables + 2 * nbytes * adds_per_window, addf))


def legs():
    L = {}

    def put(leg, form, o, how):
        floor = o["M"] * form.M + o["S"] * form.S + o["c"] * form.c
        L[leg] = {"mad_floor_per_scalar": round(floor, 1), "field_ops_per_scalar": {k: round(v, 2) for k, v in o.items()},
                  "limb_form": form.name, "mads_per_op": {"M": form.M, "S": form.S, "c": form.c}, "algorithm": how}

    # ---- the two ladders (csrc/ladder.h, fe26.h, fe28.h, fe_finish.h): rfc7748.c:156-256, the inversion shared by 32 records
    put("x25519", F10, add(times(255, LADDER_STEP), shared_inv("25519", numerators=1)),
        "255 ladder steps of 5M + 4S + 1c (rfc7748.c:186-221) + x2 / z2 under an inversion shared by 32 records")
    put("x448", F16, add(times(448, LADDER_STEP), shared_inv("448", numerators=1)),
        "448 ladder steps of 5M + 4S + 1c + the shared inversion")

    # ---- bit-exact ecn mul / mul2: the reference's formulas and window, call for call (csrc/curve.h, edwards.h, weierstrass.h)
    for C, form, nb, dbl, addf in (("ED25519", F10, 32, ED_DBL, ED_ADD), ("ED448", F16, 56, ED_DBL, ED_ADD),
                                   ("NIST256", F10, 32, WS3_DBL, WS3_ADD), ("SECP256K1", F10, 32, WS0_DBL, WS0_ADD)):
        put(C + "_ecn_mul", form, fixed_window(nb, dbl, addf), "the reference's ecn mul: table of 4 doublings + 3 additions, %d windows of 4 doublings + 1 addition" % (2 * nb))
        put(C + "_ecn_mul2", form, fixed_window(nb, dbl, addf, adds_per_window=2, tables=2),
            "two fixed-window multiplications sharing their doublings: 2 tables, %d windows of 4 doublings + 2 additions" % (2 * nb))

    # ---- fused Edwards forms (docs/curve_layer.md "Round 5")
    for C, form, steps, inv, nwin in (("ED25519", F10, 256, "25519", 65), ("ED448", F16, 448, "448", 113)):
        a1 = C == "ED448"
        MADD = A1_MADD if a1 else ED_EXT_MADD
        CACHED, CACHED_NOT, FULL = (A1_ADD_CACHED, A1_ADD_CACHED_NOT, A1_ADD) if a1 else (ED_EXT_ADD_CACHED, ED_EXT_ADD_CACHED_NOT, ED_EXT_ADD)
        lad = add(ops(M=3), shared_inv(inv), times(steps, LADDER_STEP), ops(M=12, S=1), shared_inv(inv))
        put(C + "_ecn_mul_get_fused", form, lad, "ed26l.h / ed28l.h: prep 3M, (u, w) of P under a shared inversion, %d ladder steps, Okeya-Sakurai recovery and map back 12M + 1S, "
            "(x, y) under a shared inversion" % steps)
        put(C + "_ecn_mulgen2_get_fused", form, add(lad, ops(M=1), times(nwin, MADD)),
            "f Q by the ladder form with T recovered (+1M), e G through the fixed-base table: %d mixed additions of %dM" % (nwin, MADD["M"]))
        put(C + "_ecn_mulgen_get_fused", form, add(times(nwin, MADD), shared_inv(inv)),
            "fixed-base table: %d signed 4-bit windows, one mixed addition (%dM) each, no doublings; export under a shared inversion" % (nwin, MADD["M"]))
        table = add(times(4, ED_EXT_DBL_T), times(3, FULL), ops(M=8))                  # 2P..8P, then the cached form (d T) of eight entries
        straus = add(times(2, table), times(steps - nwin, ED_EXT_DBL), times(nwin, ED_EXT_DBL_T), times(nwin, add(CACHED, CACHED_NOT)), shared_inv(inv))
        put(C + "_ecn_mul2_get_fused", form, straus, "ed26s.h / ed28s.h Straus: two projective cached tables {1..8}P, {1..8}Q, %d shared doublings, %d windows of two additions "
            "(8M + 7M), export under a shared inversion" % (steps, nwin))

    # ---- fused P-256 (wj26.h, wn_affine.h, wn_export.h): Jacobian doublings, affine window tables, mixed additions
    aff_table = add(times(4, JAC_DBL), times(3, JAC_ADD), times(8, ops(M=6, S=1)), times(1.0 / 4, INV["p256"]))   # P..8P, to Z = 1: 8 entries under 1/4 inversion
    last = add(WS3_MADD_COMPLETE, ops(M=2, S=1))                                       # the last addition is the complete mixed one on homogeneous coordinates
    export_p = shared_inv("p256")
    p_mul = add(aff_table, times(64, add(times(4, JAC_DBL), JAC_MADD)), last, export_p)
    put("NIST256_ecn_mul_get_fused", F10, p_mul, "affine table {1..8}P (4 Jacobian doublings, 3 additions, 8 entries to Z = 1 under a quarter of an inversion), 64 windows of "
        "4 doublings (3M + 5S) + 1 mixed addition (7M + 4S), the last addition complete, export under a shared inversion")
    put("NIST256_ecn_mulgen_get_fused", F10, add(times(52, JAC_MADD), export_p), "fixed-base comb, 52 five-bit windows, one Jacobian mixed addition each; shared export")
    put("NIST256_ecn_mulgen2_get_fused", F10, add(p_mul, times(52, WS3_MADD_COMPLETE)), "f Q as mul_get; e G through the fixed-base table with 52 complete mixed additions (11M + 2 m_b)")
    put("NIST256_ecn_mul2_get_fused", F10, add(times(2, aff_table), times(256, JAC_DBL), times(130, JAC_MADD), ops(M=2, S=1), export_p),
        "two affine tables, 65 windows: 256 Jacobian doublings, 130 mixed additions (wj26.h mul2_acc_aff), export")

    # ---- fused secp256k1 (glv26.h): endomorphism split, complete a = 0 formulas, one table
    ktab = add(times(4, WS0_DBL), times(3, WS0_ADD))
    export_k = shared_inv("k256")
    k_mul = add(ktab, times(33, add(times(4, WS0_DBL), times(2, WS0_ADD), ops(M=1))), export_k)
    put("SECP256K1_ecn_mul_get_fused", F10, k_mul, "k = k1 + k2 lambda: one table {1..8}P, 33 windows of 4 complete doublings (6M + 2S + 1c) + 2 complete additions (12M + 2c) + 1M "
        "by beta; shared export.  (The split's integer products, ~200 multiply-adds, are not field work and are not counted.)")
    put("SECP256K1_ecn_mulgen_get_fused", F10, add(times(52, WS0_MADD_COMPLETE), export_k), "fixed-base comb, 52 five-bit windows, one complete mixed addition (11M + 2c) each; shared export")
    put("SECP256K1_ecn_mulgen2_get_fused", F10, add(k_mul, times(52, WS0_MADD_COMPLETE)), "f Q as mul_get; e G through the fixed-base table with 52 complete mixed additions")
    put("SECP256K1_ecn_mul2_get_fused", F10, add(times(2, ktab), times(33, add(times(4, WS0_DBL), times(4, WS0_ADD), ops(M=2))), export_k),
        "both scalars split: two tables, 33 windows of 4 doublings + 4 additions + 2M by beta; shared export")
    return L


def main():
    doc = {"made_by": "tools/mad_floor.py", "definition": "field operations of the leg's algorithm x schoolbook column products of its limb form (Karatsuba where the form has it); "
           "reduction, fold and carry work is above the floor by definition", "legs": legs()}
    out = os.path.join(ROOT, "profiles", "mad_floor.json")
    json.dump(doc, open(out, "w"), indent=1)
    meas = {}
    for name in ("r06_valu_pmc.json", "r05_valu_pmc.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            meas = json.load(open(p))["legs"]
            break
    print("%-34s %10s %10s %7s   M / S / c per record" % ("leg", "floor", "kernel", "k/floor"))
    bad = 0
    for leg, e in doc["legs"].items():
        k = (meas.get(leg) or {}).get("mad_per_scalar")
        o = e["field_ops_per_scalar"]
        print("%-34s %10.0f %10s %7s   %.0f / %.0f / %.0f" % (leg, e["mad_floor_per_scalar"], "%.0f" % k if k else "-", "%.3f" % (k / e["mad_floor_per_scalar"]) if k else "-", o["M"], o["S"], o["c"]))
        bad += bool(k and k < 0.98 * e["mad_floor_per_scalar"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
