"""Emit the per-prime constant header the HIP kernels are compiled against.

`python -m modarith_amd.emit` (also run by __graft_entry__.build()) writes
modarith_amd/csrc/generated/params_<PRIME>.h for every built prime.  This is the counterpart of the
reference generators printing literal constants into field.c (e.g. the macro block
pseudo.py:1388-1411, the nres constant monty.py:1391-1394, the prime limbs in caddp/addp/subp): here
the constants go into a `struct P_<PRIME>` of constexpr members consumed by csrc/field.h, and the
kernels themselves are hand-written templates.

The progenitor addition chain (x -> x^PE, used by modpro/modinv/modsqrt; reference: external
`addchain` tool, pseudo.py:1582-1587 + 758-785) is produced by `addition_chain()` below: runs of ones
built by a doubling ladder, then stitched together with squarings.
"""
from __future__ import annotations

import os
from typing import List, Tuple

from .params import FieldParams, derive

HERE = os.path.dirname(os.path.abspath(__file__))
GEN_DIR = os.path.join(HERE, "csrc", "generated")
CORE_PRIMES = ("X25519", "NIST256", "X448")            # BASELINE.json configs; their capi_<P>.hip are hand-written
EXTRA_PRIMES = ("NIST521", "PM266", "PM383", "NUMS256W", "NIST384", "NIST224", "SECP256K1M", "NIST256Q", "ED25519Q", "ED448Q",
                "C2065", "PM336", "PM512", "GM270", "GM240", "GM360", "GM480", "GM384", "GM512", "TWEEDLE", "SIDH434", "SIDH503",
                "SECP256K1", "C41417", "ED248", "ED376", "ED500",
                "SIDH610", "SIDH751", "MFP4", "MFP7", "MFP1973", "CSIDH512", "GM378",
                "PM383M", "PM266M", "PM336M", "C41417M", "PM512M", "M607")
BUILT_PRIMES = CORE_PRIMES + EXTRA_PRIMES


# ------------------------------------------------------------------ addition chain
def _runs(e: int) -> List[Tuple[int, int]]:
    """binary expansion of e from the top as [(ones, zeros_after), ...]."""
    bits = bin(e)[2:]
    out, i = [], 0
    while i < len(bits):
        j = i
        while j < len(bits) and bits[j] == "1":
            j += 1
        k = j
        while k < len(bits) and bits[k] == "0":
            k += 1
        out.append((j - i, k - j))
        i = k
    return out


def _ones_plan(r: int, have: dict, steps: list):
    """make x^(2^r-1) available as register name have[r]; binary ladder on r, memoised."""
    if r in have:
        return have[r]
    if r % 2 == 0:
        h = _ones_plan(r // 2, have, steps)
        name = "o%d" % r
        steps.append(("dbl", name, h, r // 2))       # name = h^(2^(r/2)) * h
    else:
        h = _ones_plan(r - 1, have, steps)
        name = "o%d" % r
        steps.append(("inc", name, h))               # name = h^2 * x
    have[r] = name
    return name


def addition_chain(e: int):
    """straight-line program computing x^e: list of steps
       ("dbl", dst, src, k): dst = src^(2^k) * src      ("inc", dst, src): dst = src^2 * x
       ("start", src): acc = src                         ("run", k, src): acc = acc^(2^k) * src
       ("sqr", k): acc = acc^(2^k)                       ("mulx", dst, src): dst = src * x
    The leading run of ones is built by a doubling ladder (those squarings ARE the main chain); every
    later run is stitched from the ladder's by-products, largest first, so it costs multiplications
    only (e.g. NIST256: 253 squarings + 12 multiplications)."""
    runs = _runs(e)
    have = {1: "x"}
    steps: list = []
    # e = A*2^r + (2^r - 1) with a short head A and a long trailing run (the c*2^k - 1 primes of ED248 / ED376 / ED500):
    # with L = x^(2^r - 1) from the ladder, y = L*x = x^(2^r) and x^e = y^A * L -- the ladder's squarings are again the
    # main chain (e.g. ED248: 247 squarings + 13 multiplications)
    alt = None
    if len(runs) >= 2 and runs[-1][1] == 0 and runs[-1][0] > runs[0][0]:
        r = runs[-1][0]
        have2, steps2 = {1: "x"}, []
        L = _ones_plan(r, have2, steps2)
        alt = list(steps2)
        alt.append(("mulx", "y", L))                  # y = L * x
        alt.append(("start", "y"))
        for bit in bin(e >> r)[3:]:
            alt.append(("run", 1, "y") if bit == "1" else ("sqr", 1))
        alt.append(("run", 0, L))                     # acc = acc * L
    _ones_plan(runs[0][0], have, steps)
    prog = list(steps)
    prog.append(("start", have[runs[0][0]]))
    pending = runs[0][1]
    for ones, zeros in runs[1:]:
        left = ones
        while left:
            k = max(r for r in have if r <= left)
            prog.append(("run", pending + k, have[k]))
            pending = 0
            left -= k
        pending = zeros
    if pending:
        prog.append(("sqr", pending))
    if alt is not None:
        cost = lambda pr: chain_cost(pr)[0] + 1.4 * chain_cost(pr)[1]
        if cost(alt) < cost(prog):
            return alt
    return prog


def trailing_plan(e: int):
    """(r, A) if addition_chain(e) chose the trailing-run plan e = A*2^r + (2^r - 1), else None"""
    prog = addition_chain(e)
    if any(st[0] == "mulx" for st in prog):
        r = _runs(e)[-1][0]
        return r, e >> r
    return None


def _trailing_cpp(r: int, A: int, N: int) -> str:
    """rolled form of the trailing-run plan: the doubling ladder for L = x^(2^r - 1) as a loop over the bits of r (one
    copy of the multiplication and of the squaring loop in the instruction stream instead of one per ladder step),
    then y = L*x, y^A by the binary method over the few bits of A, times L.  Same operation count as the chain."""
    rb = r.bit_length()
    ab = A.bit_length()
    return "\n".join([
        "        spint x[%d], L[%d], t[%d], acc[%d];" % (N, N, N, N),
        "        F::modcpy(w, x);",
        "        F::modcpy(x, L);",
        "        int len = 1;",
        "#pragma unroll 1",
        "        for (int b = %d; b >= 0; b--) {                  // L = x^(2^len - 1), len -> %d" % (rb - 2, r),
        "            F::modcpy(L, t); F::modnsqr(L, len); F::modmul(L, t, L);",
        "            len *= 2;",
        "            if ((%du >> b) & 1u) { F::modsqr(L, L); F::modmul(L, x, L); len += 1; }" % r,
        "        }",
        "        F::modmul(L, x, t);                              // y = x^(2^%d)" % r,
        "        F::modcpy(t, acc);",
        "#pragma unroll 1",
        "        for (int b = %d; b >= 0; b--) {                  // y^A by the binary method, A = %s" % (ab - 2, hex(A)),
        "            unsigned long long aw = 0;",
        "            switch (b >> 6) { %s }" % " ".join("case %d: aw = 0x%xull; break;" % (i, (A >> (64 * i)) & ((1 << 64) - 1))
                                                   for i in range((ab + 63) // 64)),
        "            F::modsqr(acc, acc);",
        "            if ((aw >> (b & 63)) & 1ull) F::modmul(acc, t, acc);",
        "        }",
        "        F::modmul(acc, L, z);"])


def chain_cost(prog):
    s = m = 0
    for st in prog:
        if st[0] == "dbl":
            s += st[3]; m += 1
        elif st[0] == "inc":
            s += 1; m += 1
        elif st[0] == "mulx":
            m += 1
        elif st[0] == "run":
            s += st[1]; m += 1
        elif st[0] == "sqr":
            s += st[1]
    return s, m


def eval_chain(prog, x: int, p: int) -> int:
    """big-integer evaluation of a chain (used by tests to prove it computes x^e)."""
    reg = {"x": x % p}
    acc = None
    for st in prog:
        if st[0] == "dbl":
            reg[st[1]] = pow(reg[st[2]], 1 << st[3], p) * reg[st[2]] % p
        elif st[0] == "inc":
            reg[st[1]] = reg[st[2]] ** 2 * reg["x"] % p
        elif st[0] == "mulx":
            reg[st[1]] = reg[st[2]] * reg["x"] % p
        elif st[0] == "start":
            acc = reg[st[1]]
        elif st[0] == "run":
            acc = pow(acc, 1 << st[1], p) * reg[st[2]] % p
        elif st[0] == "sqr":
            acc = pow(acc, 1 << st[1], p)
    return acc


MAX_GENERATED_LIMBS = 16  # generator mode (modarith_amd.generate): limbs stay in VGPRs; the widest built-in field has 13
MAX_UNROLLED_MULS = 24   # beyond this the progenitor is a square-and-multiply loop (general primes)


def _binary_cpp(pe: int, N: int) -> str:
    """square-and-multiply over the bits of PE, for exponents without long runs (group orders and other
    general primes): one copy of modsqr and one of modmul in the instruction stream; the exponent is a
    public constant, so the branch is uniform across the wave"""
    nb = pe.bit_length()
    words = [(pe >> (64 * i)) & ((1 << 64) - 1) for i in range((nb + 63) // 64)]
    sw = " ".join("case %d: w = 0x%xull; break;" % (i, v) for i, v in enumerate(words))
    return "\n".join([
        "        spint x[%d], acc[%d];" % (N, N),
        "        F::modcpy(w_, x);",
        "        F::modcpy(w_, acc);",
        "#pragma unroll 1",
        "        for (int i = %d; i >= 0; i--) {" % (nb - 2),
        "            unsigned long long w = 0;",
        "            switch (i >> 6) { %s }" % sw,
        "            F::modsqr(acc, acc);",
        "            if ((w >> (i & 63)) & 1) F::modmul(acc, x, acc);",
        "        }",
        "        F::modcpy(acc, z);"])


def _chain_cpp(prog, N: int) -> str:
    lines = ["        spint x[%d], acc[%d];" % (N, N), "        F::modcpy(w, x);"]
    declared = set()
    for st in prog:
        if st[0] in ("dbl", "inc", "mulx"):
            dst = st[1]
            if dst not in declared:
                lines.append("        spint %s[%d];" % (dst, N))
                declared.add(dst)
        if st[0] == "dbl":
            _, dst, src, k = st
            lines.append("        F::modcpy(%s, %s); F::modnsqr(%s, %d); F::modmul(%s, %s, %s);" % (src, dst, dst, k, dst, src, dst))
        elif st[0] == "inc":
            _, dst, src = st
            lines.append("        F::modsqr(%s, %s); F::modmul(%s, x, %s);" % (src, dst, dst, dst))
        elif st[0] == "mulx":
            _, dst, src = st
            lines.append("        F::modmul(%s, x, %s);" % (src, dst))
        elif st[0] == "start":
            lines.append("        F::modcpy(%s, acc);" % st[1])
        elif st[0] == "run":
            lines.append("        F::modnsqr(acc, %d); F::modmul(acc, %s, acc);" % (st[1], st[2]))
        elif st[0] == "sqr":
            lines.append("        F::modnsqr(acc, %d);" % st[1])
    lines.append("        F::modcpy(acc, z);")
    return "\n".join(lines)


# ------------------------------------------------------------------ header text
def _hexu(v: int) -> str:
    return "0x%xull" % v


def _switch(name: str, ctype: str, values, fmt) -> str:
    body = " ".join("case %d: return %s;" % (i, fmt(v)) for i, v in enumerate(values))
    return "    static constexpr %s %s(int i) { switch (i) { %s default: return 0; } }" % (ctype, name, body)


def split_point(fp: FieldParams) -> int:
    """Bit position H at which csrc/field.h's FAST product path cuts operands into 32-bit halves, or 0 if the
    three 64-bit column accumulators (s0 + s1*2^H + s2*2^2H) cannot be proven overflow-free for this prime.
    Contract: every limb < 2^(radix+2) (tight limbs, [p,2p) with the top limb unmasked, generic=False sums);
    pre-multiplied operands (ma = mm*a, ta = 2a) are wider by bits(mm) / 1 bit.  n = most products per column."""
    W = fp.radix + 2
    if fp.pm or fp.bad_overflow:
        return 0                # monty.py's PM form / pseudo.py's bad_overflow form: exact products only
    if fp.family == "pseudo":
        wa = W + (fp.mm.bit_length() if fp.epm else 0)
        wb = W + (1 if fp.epm else 0)
        n = fp.nlimbs
    else:
        wa, wb, n = W, W, 2 * fp.nlimbs
    lim = 1 << 64

    def cut(n):
        best, best_cost = 0, None
        for H in range(20, 33):
            if wa - H > 32 or wb - H > 32:
                continue
            s0, s1, s2 = n << (2 * H), n * ((1 << wa) + (1 << wb)), n << max(wa + wb - 2 * H, 0)
            if s0 < lim and s1 < lim and s2 < lim and (best_cost is None or max(s0, s2) < best_cost):
                best, best_cost = H, max(s0, s2)
        return best

    best = cut(n)
    if best == 0 and fp.family != "pseudo" and not sparse_terms(fp) is None:
        # a sparse prime: a column holds the N products and ONE multiply-accumulate per non-zero prime limb above the low one
        # (field.h monty_reduce: limbs 0, +-1 and powers of two never enter the accumulators), not N of them -- ED500 = 9 x 57 bits
        # with a single such limb: ten terms of 2^59 x 2^59 instead of eighteen (round 4)
        best = cut(sparse_terms(fp))
    return best


def split_is_sparse(fp: FieldParams) -> bool:
    """True when split_point(fp) is provable only with the sparse term count: the dense count 2N has no cut.  field.h refuses to
    combine such a SPLIT with the product forms whose columns hold the dense number of accumulator terms (chain / half-limb forms)."""
    if split_point(fp) == 0 or fp.family == "pseudo" or sparse_terms(fp) is None:
        return False
    W, n, lim = fp.radix + 2, 2 * fp.nlimbs, 1 << 64
    for H in range(20, 33):
        if W - H > 32:
            continue
        if (n << (2 * H)) < lim and n * ((1 << W) + (1 << W)) < lim and (n << max(2 * W - 2 * H, 0)) < lim:
            return False            # the dense count has a cut too
    return True


def sparse_terms(fp: FieldParams):
    """products + accumulator-borne reduction terms of one column for a Montgomery prime with ndash == 1 and few non-zero limbs
    (+ 1 for the doubled cross terms of the squarings' odd term count), or None where the dense count 2N has to stand"""
    if fp.family == "pseudo" or fp.pm or fp.ndash != 1:
        return None
    k = sum(1 for i, v in enumerate(fp.ppw) if i > 0 and v not in (0, 1, -1) and (v & (v - 1)) != 0)
    return fp.nlimbs + k + 1


def chain_ok(fp: FieldParams) -> bool:
    """True if csrc/field.h may run the product loops of this prime on the 64-bit column chain (Wide::Acc):
    t = c + s0 + s1*2^H + s2*2^2H with digit() = (t & mask, t >> Radix) in one-word arithmetic.  Proven here,
    under split_point's limb contract (every limb < 2^(radix+2)): Radix - H <= 32, Radix <= 2H < 64, and the
    one-word sum c + s0 + 2^Radix stays below 2^64, where c <= max(t) >> Radix plus the one-word terms added
    in that column.  Column maxima: pseudo-Mersenne (EPM form only) N products of a (mm*a | a) by a (b | 2a)
    limb; Montgomery N products a*b plus N products digit * prime limb (both < 2^Radix) plus N+2 one-word
    terms below 2^(Radix+1)."""
    H = split_point(fp)
    R, N, W = fp.radix, fp.nlimbs, fp.radix + 2
    if H == 0 or R - H > 32 or 2 * H < R or 2 * H >= 64:
        return False
    if fp.family != "pseudo" and (2 * N) * ((1 << W) + (1 << W)) >= 1 << 64:
        return False            # SPLIT came from the sparse count (split_point): the chain's bounds below assume the dense one
    if fp.family == "pseudo":
        if not fp.epm or fp.overflow:
            return False
        col = N * (((1 << W) * fp.mm) * (1 << (W + 1)))
        words = 0
        s0 = N << (2 * H)
    else:
        col = N * (1 << W) ** 2 + (N + 1) * (1 << R) ** 2     # + the digit * p0 product of a full reduction
        words = (N + 2) << (R + 1)
        n = 2 * N + 1
        s0 = n << (2 * H)
        if n << (W + 1) >= 1 << 64 or n << (2 * max(W - H, 0)) >= 1 << 64:   # s1, s2 with that extra product
            return False
    # carry fixed point: c = (col + words + c) >> R
    c = 0
    for _ in range(4):
        c = ((col + words + c) >> R) + 2
    return c + words + s0 + (1 << R) < (1 << 64)


def header_text(fp: FieldParams) -> str:
    N = fp.nlimbs
    prog = addition_chain(fp.pe)
    sq, mu = chain_cost(prog)
    L = []
    L.append("// GENERATED by modarith_amd/emit.py from modarith_amd/params.py -- do not edit.")
    L.append("// prime %s = %s, family %s" % (fp.name, hex(fp.p), fp.family))
    L.append("#pragma once")
    L.append('#include "../field.h"')
    L.append("namespace ma {")
    L.append("struct P_%s {" % fp.name)
    L.append('    static constexpr const char* NAME = "%s";' % fp.name)
    L.append("    static constexpr int N = %d, RADIX = %d, NBITS = %d, NBYTES = %d, XCESS = %d, PM1D2 = %d;"
             % (N, fp.radix, fp.n, fp.nbytes, fp.xcess, fp.pm1d2))
    L.append("    static constexpr bool MONTGOMERY = %s;" % ("true" if fp.montgomery else "false"))
    L.append("    static constexpr int SPLIT = %d;   // FAST product path: operand cut position, 0 = not provable (emit.split_point)" % split_point(fp))
    L.append("    static constexpr bool SPLIT_SPARSE = %s;   // SPLIT proven from the SPARSE term count (emit.sparse_terms): holds for Field::monty_mul / monty_reduce only" % ("true" if split_is_sparse(fp) else "false"))
    L.append("    static constexpr bool CHAIN = %s;   // FAST product loops on the 64-bit column chain (emit.chain_ok)" % ("true" if chain_ok(fp) else "false"))
    # pseudo-Mersenne block (dummies for Montgomery primes)
    L.append("    static constexpr unsigned long long M = %s, MM = %s;" % (_hexu(fp.m if not fp.montgomery else 0), _hexu(fp.mm)))
    L.append("    static constexpr bool OVERFLOW = %s, FRED = %s, EPM = %s, CARRY_ON = %s, BAD_OVERFLOW = %s;"
             % tuple("true" if b else "false" for b in (fp.overflow, fp.fred, fp.epm, fp.carry_on, fp.bad_overflow)))
    # Montgomery block (dummies for pseudo-Mersenne primes)
    neg = [i for i, v in enumerate(fp.ppw) if i > 0 and v == -1]
    L.append("    static constexpr bool E = %s;" % ("true" if fp.E else "false"))
    L.append("    static constexpr unsigned long long PM_M = %s;   // monty.py's PM form: ppw(0) = -PM_M (0 = not that form)" % _hexu(fp.m if fp.pm else 0))
    L.append("    static constexpr unsigned long long NDASH = %s;" % _hexu(fp.ndash))
    L.append("    static constexpr int TRIN = %d, NEG_LIMB = %d;" % (fp.trin, neg[0] if neg else 0))
    br = (1 << (fp.n + fp.radix)) // fp.p if fp.montgomery else 0
    L.append("    static constexpr unsigned long long BARRETT_R = %s;  // floor(2^(n+Radix)/p) (monty.py:923)" % _hexu(br if br < 1 << 64 else 0))
    L.append("    static constexpr int BARRETT_SHIFT = %d;                 // (n-64) %% Radix (monty.py:930)" % ((fp.n - 64) % fp.radix))
    ppw = fp.ppw if fp.ppw else [0]
    L.append(_switch("ppw", "long long", ppw, lambda v: "%dll" % v if (abs(v) < 10 or v < 0) else ("0x%xll" % v)))
    r2 = fp.r2 if fp.r2 else [0] * N
    L.append(_switch("r2", "unsigned long long", r2, _hexu))
    # non-zero prime limbs for caddp/addp/subp
    L.append("    static constexpr int PP_CNT = %d;" % len(fp.pp))
    L.append(_switch("pp_idx", "int", [t[0] for t in fp.pp], str))
    L.append(_switch("pp_sgn", "int", [t[1] for t in fp.pp], str))
    L.append(_switch("pp_val", "unsigned long long", [t[2] for t in fp.pp], _hexu))
    L.append(_switch("roi", "unsigned long long", fp.roi, _hexu))
    tp = trailing_plan(fp.pe)
    if tp is not None:
        L.append("    // progenitor for PE = %s = A*2^r + (2^r - 1), r = %d, A = %s: rolled doubling ladder, %d squarings + %d multiplications"
                 % (hex(fp.pe), tp[0], hex(tp[1]), sq, mu))
        L.append("    template <class F>")
        L.append("    static __device__ __forceinline__ void modpro_chain(const spint* w, spint* z) {")
        L.append(_trailing_cpp(tp[0], tp[1], N))
        L.append("    }")
    elif mu <= MAX_UNROLLED_MULS:
        L.append("    // progenitor chain for PE = %s: %d squarings + %d multiplications" % (hex(fp.pe), sq, mu))
        L.append("    template <class F>")
        L.append("    static __device__ __forceinline__ void modpro_chain(const spint* w, spint* z) {")
        L.append(_chain_cpp(prog, N))
        L.append("    }")
    else:
        L.append("    // progenitor for PE = %s: no run structure, square-and-multiply loop (%d squarings + %d multiplications)"
                 % (hex(fp.pe), fp.pe.bit_length() - 1, bin(fp.pe).count("1") - 1))
        L.append("    template <class F>")
        L.append("    static __device__ __forceinline__ void modpro_chain(const spint* w_, spint* z) {")
        L.append(_binary_cpp(fp.pe, N))
        L.append("    }")
    L.append("};")
    L.append("}  // namespace ma")
    return "\n".join(L) + "\n"


BUILT_CURVES = ("ED25519", "ED448", "NUMS256E", "ED248", "ED376", "ED500")


def curve_header_text(name: str) -> str:
    """constants of one Edwards curve as `struct C_<NAME>` (counterpart of curve.py's curve.c: COF,
    CONSTANT_A, CONSTANT_B or constant_b[], constant_x[], constant_y[]; curve.py:244-298)."""
    from .curves import curve
    return curve_header_text_of(curve(name))


def curve_header_text_of(c) -> str:
    """the same for any EdwardsCurve object with its field parameters attached (c.fp): the built-in table, or a curve of the
    caller's own (modarith_amd.generate.generate_curve, the counterpart of curve.py's "Insert your own!")"""
    N = c.fp.nlimbs
    L = ["// GENERATED by modarith_amd/emit.py from modarith_amd/curves.py -- do not edit.",
         "// Edwards curve %s: %d*x^2 + y^2 = 1 + d*x^2*y^2 over the %s field" % (c.name, c.a, c.field),
         "#pragma once",
         '#include "params_%s.h"' % c.field,
         "namespace ma {",
         "struct C_%s {" % c.name,
         "    using FieldParams = P_%s;" % c.field,
         "    static constexpr int A = %d, COF = %d;" % (c.a, c.cof),
         "    static constexpr bool B_SMALL = %s;" % ("true" if c.small_b else "false"),
         "    static constexpr int B_INT = %d;       // CONSTANT_B when small (curve.py:256-257)" % (c.d if c.small_b else 0),
         "    static constexpr int SMALL_X = %d;     // CONSTANT_X when the generator is given by a small x (curve.py:239-240), else 0" % (c.gx if c.small_x else 0)]
    bl = c.internal(c.d) if not c.small_b else [0] * N
    L.append(_switch("b", "unsigned long long", bl, _hexu))
    L.append(_switch("gx", "unsigned long long", c.internal(c.gx), _hexu))
    L.append(_switch("gy", "unsigned long long", c.internal(c.gy), _hexu))
    L += ["};", "}  // namespace ma"]
    return "\n".join(L) + "\n"


def capi_unit_text(name: str) -> str:
    """translation unit with the C-ABI entry points of one field-only prime"""
    return ("// GENERATED by modarith_amd/emit.py -- do not edit.\n"
            "// C-ABI entry points for %s (field only); body: ../capi_prime.inc\n"
            '#include "params_%s.h"\n#define MA_P ma::P_%s\n#define MA_NAME %s\n#include "../capi_prime.inc"\n' % (name, name, name, name))


def field_table_text(primes=BUILT_PRIMES) -> str:
    """rows of modarith_amd_field_info(): the macro block of each prime's field.c (pseudo.py:1403-1407)"""
    rows = []
    for name in primes:
        fp = derive(name)
        rows.append('    {"%s", %d, %d, %d, %d, %d},' % (name, fp.nlimbs, fp.radix, fp.n, fp.nbytes, 1 if fp.montgomery else 0))
    return "// GENERATED by modarith_amd/emit.py -- do not edit.\n" + "\n".join(rows) + "\n"


def _write(path: str, text: str) -> str:
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


BUILT_WCURVES = ("NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W")


def wcurve_header_text(name: str) -> str:
    """constants of one short-Weierstrass curve (curve.py's curve.c: CONSTANT_A, constant_b[], constant_b3[],
    constant_x[], constant_y[]; curve.py:244-298)"""
    from .curves import wcurve
    return wcurve_header_text_of(wcurve(name))


def wcurve_header_text_of(c) -> str:
    L = ["// GENERATED by modarith_amd/emit.py from modarith_amd/curves.py -- do not edit.",
         "// Weierstrass curve %s: y^2 = x^3 %+d*x + b over the %s field" % (c.name, c.a, c.field),
         "#pragma once",
         '#include "params_%s.h"' % c.field,
         "namespace ma {",
         "struct C_%s {" % c.name,
         "    using FieldParams = P_%s;" % c.field,
         "    static constexpr int A = %d;" % c.a,
         "    static constexpr int SMALL_B = %d;   // curve.py's CONSTANT_B when |b| < 2^28, else 0 (b, b3 below are used)" % (c.b if c.small_b else 0),
         "    static constexpr int SMALL_X = %d;   // curve.py's CONSTANT_X when the generator is given by a small x, else 0" % (c.gx if c.small_x else 0),
         _switch("b", "unsigned long long", c.internal(c.b), _hexu),
         _switch("b3", "unsigned long long", c.internal(3 * c.b), _hexu),
         _switch("gx", "unsigned long long", c.internal(c.gx), _hexu),
         _switch("gy", "unsigned long long", c.internal(c.gy), _hexu),
         "};", "}  // namespace ma"]
    return "\n".join(L) + "\n"


COMB_CURVES = {"NIST256": (286, 5), "SECP256K1": (0, 5)}      # curve -> (log2 of the Montgomery factor of the fused kernels' field form, window width)


def comb_header_text(name: str) -> str:
    """fixed-base table of the fused generator multiplication (csrc/wn26.h wn26_mulgen_acc): for every W-bit window i
    the affine multiples m * 2^(W i) * G, m = 1..2^(W-1), coordinates as ten 26-bit limbs of the fused kernels' field form
    (value * 2^286 mod p for P-256, the plain value for secp256k1).  Plain integer curve arithmetic on the constants of
    curves.py (curve.py:157-198)."""
    from .curves import wcurve
    c = wcurve(name)
    p, a = c.fp.p, c.a
    shift, W = COMB_CURVES[name]
    windows = -(-(8 * c.fp.nbytes + 1) // W)

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1 + a) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    limbs = lambda v: [((v << shift) % p >> (26 * k)) & ((1 << 26) - 1) for k in range(10)]
    rows = []
    B = (c.gx, c.gy)
    for i in range(windows):
        T = None
        for m in range(1, (1 << (W - 1)) + 1):
            T = add(T, B)
            rows.append("    " + ", ".join("0x%x" % v for v in limbs(T[0]) + limbs(T[1])) + ",   /* %d * 2^%d * G */ \\" % (m, W * i))
        for _ in range(W):
            B = add(B, B)
    L = ["// GENERATED by modarith_amd/emit.py from modarith_amd/curves.py -- do not edit.",
         "// Fixed-base table of %s for ecn_%s_mulgen_get_batch: [window 0..%d][multiple 1..%d][x, y][10 limbs of 26 bits]," % (
             name, name.lower(), windows - 1, 1 << (W - 1)),
         "// coordinates in the field form of the fused kernels (%s)." % ("value * 2^%d mod p" % shift if shift else "plain value"),
         "#pragma once",
         "#define COMB_%s_W %d" % (name, W),
         "#define COMB_%s_WINDOWS %d" % (name, windows),
         "#define COMB_%s_VALUES \\" % name] + rows + ["    /* end */", ""]
    return "\n".join(L)


COMB_EDWARDS = {"ED25519": (4, "fe26"), "ED448": (4, "fe28")}      # curve -> (window width, limb form)


def comb_edwards_header_text(name: str) -> str:
    """fixed-base table of the fused generator multiplication on the Edwards curves (csrc/ed26.h / ed28.h *_mulgen_get_one):
    for every W-bit window i the affine multiples m * 2^(W i) * G, m = 1..2^(W-1), in the cached form of the mixed addition --
    ED25519: (y+x, y-x, 2dxy) as ten 25.5-bit limbs (fe26.h), ED448: (x, y, 39081 x y) as sixteen 28-bit limbs (fe28.h).
    Plain integer curve arithmetic on the constants of curves.py (curve.py:85-105)."""
    from .curves import curve
    c = curve(name)
    p, a, d = c.fp.p, c.a, c.d % c.fp.p
    W, form = COMB_EDWARDS[name]
    windows = -(-(8 * c.fp.nbytes + 1) // W)

    def add(P, Q):
        (x1, y1), (x2, y2) = P, Q
        t = d * x1 * x2 * y1 * y2 % p
        return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, p) % p, (y1 * y2 - a * x1 * x2) * pow(1 - t, -1, p) % p)

    if form == "fe26":
        pos = lambda i: (51 * i + 1) // 2
        limbs = lambda v: [(v >> pos(i)) & ((1 << (25 if i & 1 else 26)) - 1) for i in range(10)]
        cached = lambda x, y: limbs((y + x) % p) + limbs((y - x) % p) + limbs(2 * d * x * y % p)
    else:
        limbs = lambda v: [(v >> (28 * i)) & ((1 << 28) - 1) for i in range(16)]
        cached = lambda x, y: limbs(x) + limbs(y) + limbs((-d) % p * x * y % p)
    rows = []
    B = (c.gx, c.gy)
    for i in range(windows):
        T = (0, 1)
        for m in range(1, (1 << (W - 1)) + 1):
            T = add(T, B)
            rows.append("    " + ", ".join("0x%x" % v for v in cached(*T)) + ",   /* %d * 2^%d * G */ \\" % (m, W * i))
        for _ in range(W):
            B = add(B, B)
    L = ["// GENERATED by modarith_amd/emit.py from modarith_amd/curves.py -- do not edit.",
         "// Fixed-base table of %s for ecn_%s_mulgen_get_batch: [window 0..%d][multiple 1..%d][%s][limbs]." % (
             name, name.lower(), windows - 1, 1 << (W - 1), "y+x, y-x, 2dxy: 10 limbs of 25.5 bits" if form == "fe26" else "x, y, 39081xy: 16 limbs of 28 bits"),
         "#pragma once",
         "#define COMB_%s_W %d" % (name, W),
         "#define COMB_%s_WINDOWS %d" % (name, windows),
         "#define COMB_%s_VALUES \\" % name] + rows + ["    /* end */", ""]
    return "\n".join(L)


# the 32 function names of a generated field.c in emitted order (pseudo.py:1413-1445 functions(), monty.py:1885-1917)
FIELD_C_NAMES = ("prop", "flatten", "modfsb", "modadd", "modsub", "modneg", "modmli", "modmul", "modsqr", "modcpy", "modnsqr", "modpro", "modinv",
                 "nres", "redc", "modis1", "modis0", "modzer", "modone", "modint", "modqr", "modcmv", "modcsw", "modsqrt", "modshl", "modshr",
                 "modhaf", "mod2r", "modexp", "modimp", "modsign", "modcmp")
INCLUDE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


def field_shim_text(fp: FieldParams, tag: str = None) -> str:
    """include/field_<PRIME>.h: what a consumer includes where the reference says "paste field.c here" (rfc7748.c:24-28,
    edwards.c:19-23 @field@, weierstrass.c:16-20, edge.c:5-9): the macro block at the top of the generated field.c
    (pseudo.py:1388-1411, monty.py:1859-1882) with this driver's values, and the 32 undecorated names of pseudo.py:1413-1445
    mapped onto the library's <fn>_<PRIME>_ct entry points (the names decoration=True gives them, pseudo.py:1940-1944)."""
    tag = tag or fp.name
    L = ["/* include/field_%s.h -- EMITTED by modarith_amd/emit.py field_shim_text(); do not edit." % tag,
         " *",
         " * Put  #include \"field_%s.h\"  where the reference's templates say \"paste field.c here\" (rfc7748.c:24-28," % tag,
         " * edwards.c:19-23 @field@, weierstrass.c:16-20, edge.c:5-9; automated there by curve.py:335-351) and link",
         " * libmodarith_amd.so%s: the template's calls modmul(a, b, c) ... then run" % ("" if tag in BUILT_PRIMES else " and the field's plug-in"),
         " * on the GPU one element at a time (host pointers, the reference's signatures and aliasing rules; a bring-up path --",
         " * throughput comes from the <fn>_%s_batch entry points of modarith_amd.h)." % tag,
         " * prime %s = %s, %s" % (fp.name, hex(fp.p), "monty.py form" if fp.montgomery else "pseudo.py form"),
         " */",
         "#ifndef MODARITH_AMD_FIELD_%s_H" % tag,
         "#define MODARITH_AMD_FIELD_%s_H" % tag,
         "#include <stdio.h>",
         "#include <stdint.h>",
         '#include "modarith_amd.h"',
         "MODARITH_AMD_DECLARE(%s)" % tag if tag not in BUILT_PRIMES else "/* (modarith_amd.h declares the %s entry points) */" % tag,
         "",
         "#define sspint int64_t",
         "#define spint uint64_t",
         "#define dpint __uint128_t",
         "#define sdpint __int128_t",
         "#define Wordlength 64",
         "#define Nlimbs %d" % fp.nlimbs,
         "#define Radix %d" % fp.radix,
         "#define Nbits %d" % fp.n,
         "#define Nbytes %d" % fp.nbytes,
         ""]
    if fp.montgomery:
        L.append("#define MONTGOMERY")
        if fp.name[0].isalpha():
            L.append("#define %s" % fp.name.upper())
        if fp.trin > 0:
            L.append("#define MULBYINT")
    else:
        L += ["#define MERSENNE", "#define MULBYINT"]
        if fp.name[0].isalpha():
            L.append("#define %s" % fp.name)
    L.append("")
    L += ["#define %s %s_%s_ct" % (fn, fn, tag) for fn in FIELD_C_NAMES]
    L += ["", "#endif", ""]
    return "\n".join(L)


def emit_field_shims(primes=CORE_PRIMES, out_dir: str = INCLUDE_DIR) -> List[str]:
    return [_write(os.path.join(out_dir, "field_%s.h" % name), field_shim_text(derive(name))) for name in primes]


def emit_all(primes=BUILT_PRIMES, out_dir: str = GEN_DIR) -> List[str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = [_write(os.path.join(out_dir, "field_table.inc"), field_table_text(primes))]
    for name in COMB_CURVES:
        paths.append(_write(os.path.join(out_dir, "comb_%s.h" % name), comb_header_text(name)))
    for name in COMB_EDWARDS:
        paths.append(_write(os.path.join(out_dir, "comb_%s.h" % name), comb_edwards_header_text(name)))
    for name in EXTRA_PRIMES:
        if name in primes:
            paths.append(_write(os.path.join(out_dir, "capi_%s.hip" % name), capi_unit_text(name)))
    for name in BUILT_WCURVES:
        paths.append(_write(os.path.join(out_dir, "curve_%s.h" % name), wcurve_header_text(name)))
    for name in BUILT_CURVES:
        text = curve_header_text(name)
        path = os.path.join(out_dir, "curve_%s.h" % name)
        if not os.path.exists(path) or open(path).read() != text:
            with open(path, "w") as f:
                f.write(text)
        paths.append(path)
    for name in primes:
        text = header_text(derive(name))
        path = os.path.join(out_dir, "params_%s.h" % name)
        if not os.path.exists(path) or open(path).read() != text:
            with open(path, "w") as f:
                f.write(text)
        paths.append(path)
    if out_dir == GEN_DIR:
        paths += emit_field_shims()
    return paths


if __name__ == "__main__":
    for p in emit_all():
        print(p)
