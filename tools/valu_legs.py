"""The VALU-bound legs of bench.py and the kernels each one launches : shared by tools/run_valu_legs.py (one pass of every leg, the
target of the counter passes), tools/collect_valu_legs_pmc.py (counters -> profiles/<tag>_valu_pmc.json) and, through that file only, bench.py.
A kernel entry: (substring of the demangled kernel name, object file glob under modarith_amd/build, records per pass of the kernel
body: 1 unless a lane handles several records per outer iteration, loops without a compile-time trip count: "rounds" = the shared
inversions' per-lane element count, else 1)."""

LOG2 = {"x25519": 21, "x448": 19, "ED25519": 19, "ED448": 18, "NIST256": 19, "SECP256K1": 19}


def legs():
    out = {
        "x25519": [("k_x25519_fe26_xz", "capi_X25519.o", 1, 1), ("k_fe_finish<ma::Fe26", "capi_X25519.o", 1, "rounds")],
        "x448": [("k_x448_fe28_xz", "capi_X448.o", 1, 1), ("k_fe_finish<ma::Fe28", "capi_X448.o", 1, "rounds")],
    }
    for C, obj, low, g, fam in (("ED25519", "capi_ED25519", "ed25519", 4, "Edwards"), ("ED448", "capi_ED448", "ed448", 2, "Edwards"),
                                ("NIST256", "capi_NIST256W", "nist256", 4, "Weierstrass"), ("SECP256K1", "capi_SECP256K1W", "secp256k1", 4, "Weierstrass")):
        base = "capi_" + C
        # (round 6: the fast class, GUARD = 1; the exact class behind it -- "... Field<P, false, true> >, -1>" -- only votes and leaves)
        fld = {"ED25519": "ma::FieldH51<", "ED448": "ma::FieldH56<", "NIST256": "ma::Field<ma::P_NIST256, true", "SECP256K1": "ma::Field<ma::P_SECP256K1, true"}[C]
        out[C + "_ecn_mul"] = [("k_ed_mul<ma::%s<ma::C_%s, %s" % (fam, C, fld), obj + "_part1.o", 1, 1)]
        out[C + "_ecn_mul2"] = [("k_ed_mul2<ma::%s<ma::C_%s, %s" % (fam, C, fld), obj + "_part2.o", 1, 1)]
        if C in ("ED25519", "ED448"):
            # the ladder form (csrc/ed26l.h, ed28l.h): prep, the shared inversion in front, the ladder, the shared inversion + export
            T, NW, PF = ("LadT25519", 4, "P_X25519") if C == "ED25519" else ("LadT448", 7, "P_X448")
            for leg, tag, unit, lad in (("_ecn_mul_get_fused", 1, "F.o", "k_%s_lad(" % low), ("_ecn_mulgen2_get_fused", 2, "G.o", "k_%s_lad_gen2" % low)):
                out[C + leg] = [("k_edlad_prep<ma::%s, %d>" % (T, tag), base + unit, 1, 1), ("SinkWords<%d>, %d>" % (NW, tag), base + unit, 1, "rounds"),
                                ("SinkExportBE<ma::%s>, %d>" % (PF, tag), base + unit, 1, "rounds"), (lad, base + unit, 1, 1)]
        else:
            # the window kernels of csrc/wj26.h / glv26.h and the shared inversion + export behind them (csrc/wn_export.h)
            FX = "ma::Fm26, ma::P_NIST256" if C == "NIST256" else "ma::Fk26, ma::P_SECP256K1"
            out[C + "_ecn_mul_get_fused"] = [("k_%s_mul_get" % low, base + "F.o", 1, 1), ("k_wn_export<%s, 1>" % FX, base + "F.o", 1, "rounds")]
            out[C + "_ecn_mulgen2_get_fused"] = [("k_%s_mulgen2_get" % low, base + "G.o", 1, 1), ("k_wn_export<%s, 3>" % FX, base + "G.o", 1, "rounds")]
            if C == "NIST256":
                # P-256: the window table of every record in a kernel of its own, brought to Z = 1 by a shared inversion (csrc/wn_affine.h)
                out[C + "_ecn_mul_get_fused"] += [("k_nist256_table(", base + "F.o", 1, 1), ("k_wn_table_affine<ma::Fm26, true, 1>", base + "F.o", 1, "rounds")]
                out[C + "_ecn_mulgen2_get_fused"] += [("k_nist256_table2(", base + "G.o", 1, 1), ("k_wn_table_affine<ma::Fm26, true, 3>", base + "G.o", 1, "rounds")]
        if C in ("ED25519", "ED448"):
            out[C + "_ecn_mul2_get_fused"] = [("k_%s_mul2_straus" % low, base + "F2.o", 1, 1), ("SinkExportBE<ma::%s>, 3>" % ("P_X25519" if C == "ED25519" else "P_X448"), base + "F2.o", 1, "rounds")]
        else:
            out[C + "_ecn_mul2_get_fused"] = [("k_%s_mul2_get" % low, base + "F2.o", 1, 1), ("k_wn_export<%s, 2>" % FX, base + "F2.o", 1, "rounds")]
            if C == "NIST256":
                out[C + "_ecn_mul2_get_fused"] += [("k_nist256_tables(", base + "F2.o", 1, 1), ("k_wn_table_affine<ma::Fm26, true, 2>", base + "F2.o", 1, "rounds")]
        if C in ("ED25519", "ED448"):
            out[C + "_ecn_mulgen_get_fused"] = [("k_%s_mulgen<false>" % low, base + "G.o", 1, 1),
                                                ("SinkExportBE<ma::%s>, %d>" % (("P_X25519", 4) if C == "ED25519" else ("P_X448", 3)), base + "G.o", 1, "rounds")]
        else:
            out[C + "_ecn_mulgen_get_fused"] = [("k_%s_mulgen(" % low, base + "G.o", 1, 1), ("k_wn_export<%s, 4>" % FX, base + "G.o", 1, "rounds")]
    return out


def records(leg):
    """records one pass of tools/run_valu_legs.py hands to the leg"""
    if leg in ("x25519", "x448"):
        return 1 << LOG2[leg]
    C = leg.split("_")[0]
    n = 1 << LOG2[C]
    return n // 2 if ("mul2_get" in leg or "mulgen2_get" in leg) else n
