// modarith_amd/csrc/fh56.h -- the 2^448 - 2^224 - 1 field of field.c (8 x 56-bit limbs, Montgomery form, monty.py) with every
// element RESIDENT in half-limb form: sixteen 32-bit words h[0..15], limb k = h[2k] + 2^28 h[2k+1], h[2k] < 2^28 always; h[2k+1]
// carries whatever the limb holds above bit 28 (28 bits for a masked digit, more where the reference leaves a limb unmasked: the
// top limb of a product or a sum).  The SAME element as Field<P_X448> holds, with the same limbs whenever those are below 2^56 (see
// from_limbs() for the two excess bits the API admits on input), and every function returns exactly the limbs the reference's function returns (monty.py:417-490
// modadd / modsub / modneg generic forms, 597-838 the shape-aware product of the trinomial with its virtual limb, 1386-1399
// modmli) for field elements as the API defines them (limbs below 2^58, VALUE below 2p for the sums); tools/fe_host_check.hip
// run_fh56.  csrc/fh51.h is the same idea for 2^255-19 and says why: the scalar multiplications of the curve layer are chains of
// thousands of products with as many sums between them; Field<P_X448>::monty_mul_half_tri multiplies on half limbs already, but
// cuts each 64-bit limb on the way in (and + 64-bit shift per limb and operand) and glues the digits back on the way out, and the
// sums run their carry chains in 64-bit instructions at twice the issue cost of the 32-bit ones used here.
#pragma once
#include "field.h"

namespace ma {

template <class P>
struct FieldH56 {
    using L = Field<P, true, true>;                 // the limb-form functions (same element, same limbs)
    static_assert(L::MHALF_TRI, "FieldH56 is the half-limb resident form of the 8 x 56-bit Montgomery field 2^448 - 2^224 - 1");
    static_assert(P::PP_CNT == 3 && P::pp_idx(0) == 0 && P::pp_sgn(0) < 0 && P::pp_val(0) == 1 && P::pp_idx(1) == 4 && P::pp_sgn(1) < 0 && P::pp_val(1) == 1 &&
                  P::pp_idx(2) == 7 && P::pp_sgn(2) > 0 && P::pp_val(2) == (1ull << 56), "p = 2^448 - 2^224 - 1 as limb terms");
    using limb_t = uint32_t;
    static constexpr int N = 8;                     // limbs of the element (HBM form)
    static constexpr int NL = 16;                   // resident words
    static constexpr int H = 28;
    static constexpr uint32_t HM = (1u << H) - 1u;

    // Limbs -> resident words, NORMALISED: the excess of a limb over 56 bits (the API admits two bits: limbs below 2^58) moves up
    // into the next limb, the top word keeps what is left.  The integer is unchanged, and in this field that is all that matters:
    // the Montgomery product is a function of the integer product of its operands (every column carries into the next; monty.py:
    // 597-838) and the sums are functions of the integer sum, so every function below returns the reference's limbs for the
    // reference's operand whatever the limbs of that operand were.  (Why normalise at all: with odd words of 30 bits the first carry
    // chain of modadd would reach 2^31 in a signed 32-bit word.)  For limbs below 2^56 -- every output of a field function except
    // its top limb -- this is the plain split and to_limbs() returns the limbs that came in.
    static MA_DEV void from_limbs(const spint* a, uint32_t* h) {
        uint32_t c = 0;
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            const uint32_t lo = ((uint32_t)a[k] & HM) + c;
            h[2 * k] = lo & HM;
            const uint32_t hi = (uint32_t)(a[k] >> H) + (lo >> H);
            if constexpr (k < N - 1) {
                h[2 * k + 1] = hi & HM;
                c = hi >> H;
            } else {
                h[2 * k + 1] = hi;
            }
        });
    }
    static MA_DEV void to_limbs(const uint32_t* h, spint* a) {
        static_for<0, N>([&](auto K) { a[K] = (spint)h[2 * K] + ((spint)h[2 * K + 1] << H); });
    }
    // one 64-bit word per limb for the window tables: the two halves side by side (no shifts on either side)
    static MA_DEV spint pack(const uint32_t* h, int k) { return (spint)h[2 * k] | ((spint)h[2 * k + 1] << 32); }
    static MA_DEV void unpack(spint w, uint32_t* h, int k) { h[2 * k] = (uint32_t)w; h[2 * k + 1] = (uint32_t)(w >> 32); }

    static MA_DEV void modcpy(const uint32_t* a, uint32_t* c) { static_for<0, NL>([&](auto I) { c[I] = a[I]; }); }
    static MA_DEV void modzer(uint32_t* a) { static_for<0, NL>([&](auto I) { a[I] = 0; }); }
    static MA_DEV void modone(uint32_t* a) {        // nres(1) (monty.py:1386-1399): not hot (the neutral element of a multiplication)
        spint x[N];
        L::modone(x);
        from_limbs(x, a);
    }
    static MA_DEV void modcmv(int b, const uint32_t* g, uint32_t* f) {
        const bool take = (b & 1) != 0;
        static_for<0, NL>([&](auto I) {
            const uint32_t x = g[I], y = f[I];
            f[I] = take ? x : y;
        });
    }

    // ---------------------------------------------------------------- add / sub / neg (generic=True forms)
    // The reference: n = a (+-) b limb-wise, (modadd: n -= 2p,) prop, n += 2p under the sign mask of the top limb, prop
    // (monty.py:417-490).  2p = 2^449 - 2^225 - 2: -2p is +2 on limb 0 (h[0]), +2 on limb 4 (h[8]), -2^57 on limb 7 (= -2^29 on
    // h[15]).  Both props run as 32-bit chains over the fifteen half boundaries; the first one is fused with the limb-wise sum,
    // carries are arithmetic shifts as in prop (monty.py:352-380), the top word h[15] stays unmasked as the top limb does.  The
    // same integer at every step, hence the same digits.  Inputs: |h[i]| < 2^29 (top word < 2^30): what from_limbs() and every function here leave.
    // RIPPLE = false: the "_u" forms (Field<P>::modadd_u): the closing chain is left out, h[0], h[8] and the top word carry the +2p.
    static constexpr int32_t p2(int i) { return (i == 0 || i == 8) ? 2 : 0; }
    template <bool RIPPLE = true, class First>
    static MA_DEV void chains(First first, uint32_t* n) {
        int32_t c = 0;
        static_for<0, NL - 1>([&](auto I) {
            const int32_t x = first(I) + c;
            n[I] = (uint32_t)x & HM;
            c = x >> H;
        });
        const int32_t top = first(std::integral_constant<int, NL - 1>{}) + c;
        const int32_t m = top >> 31;                         // all ones if the value is negative
        if constexpr (!RIPPLE) {
            n[0] = (uint32_t)((int32_t)n[0] - (2 & m));      // (may go negative: the consumer's first chain takes them as signed words)
            n[8] = (uint32_t)((int32_t)n[8] - (2 & m));
            n[NL - 1] = (uint32_t)(top + ((1 << 29) & m));
            return;
        }
        c = 0;
        static_for<0, NL - 1>([&](auto I) {
            constexpr int i = I;
            int32_t x = (int32_t)n[i] + c;
            if constexpr (p2(i) != 0) x -= p2(i) & m;
            n[i] = (uint32_t)x & HM;
            c = x >> H;
        });
        n[NL - 1] = (uint32_t)(top + ((1 << 29) & m) + c);
    }
    static MA_DEV void modadd(const uint32_t* a, const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains([&](auto I) -> int32_t {
            constexpr int i = I;
            return (int32_t)(a[i] + b[i]) + p2(i) - (i == NL - 1 ? (1 << 29) : 0);
        }, r);
        modcpy(r, n);
    }
    static MA_DEV void modsub(const uint32_t* a, const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains([&](auto I) -> int32_t { return (int32_t)(a[I] - b[I]); }, r);
        modcpy(r, n);
    }
    static MA_DEV void modneg(const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains([&](auto I) -> int32_t { return -(int32_t)b[I]; }, r);
        modcpy(r, n);
    }
    static MA_DEV void modadd_u(const uint32_t* a, const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains<false>([&](auto I) -> int32_t {
            constexpr int i = I;
            return (int32_t)(a[i] + b[i]) + p2(i) - (i == NL - 1 ? (1 << 29) : 0);
        }, r);
        modcpy(r, n);
    }
    static MA_DEV void modsub_u(const uint32_t* a, const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains<false>([&](auto I) -> int32_t { return (int32_t)(a[I] - b[I]); }, r);
        modcpy(r, n);
    }
    static MA_DEV void modneg_u(const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains<false>([&](auto I) -> int32_t { return -(int32_t)b[I]; }, r);
        modcpy(r, n);
    }

    // ---------------------------------------------------------------- products
    // Field<P>::monty_mul_half_tri on resident words: the 256 (136) half products go into 32 half columns with one carry chain; the
    // reference's borrow convention for the negative middle limb enters as explicit words (Q - v_0 at limb column NEG, the scratch
    // word mask - v_{C-NEG} + v_{C-N} later; monty.py:597-627, 717-738), split exactly at 2^28.  Half digits u_0 .. u_17 are the
    // Montgomery digits v_0 .. v_8; half columns 18 .. 31 are limbs 0 .. 6 of the result, and what is left is its top limb,
    // cy + v_8 - 1 (monty.py:830-838), not masked.
    template <bool SQR>
    static MA_DEV void product(const uint32_t* f, const uint32_t* g, uint32_t* c) {
        constexpr int M = NL, NEG = P::NEG_LIMB;
        uint32_t u[4 * N], f2[M];
        if constexpr (SQR) static_for<0, M>([&](auto I) { f2[I] = 2u * f[I]; });
        uint64_t cy = 0;
        static_for<0, 4 * N>([&](auto KK) {
            constexpr int k = KK, C = k / 2, h = k % 2;
            uint64_t acc = cy;
            constexpr int lo = k < M ? 0 : k - (M - 1), hi = k < M ? k : M - 1;
            if constexpr (lo <= hi) {
                static_for<lo, hi + 1>([&](auto II) {
                    constexpr int i = II, j = k - i;
                    if constexpr (!SQR) {
                        acc += (uint64_t)f[i] * g[j];
                        MA_PIN(acc);
                    } else if constexpr (i <= j) {
                        acc += (uint64_t)((i < j) ? f2[i] : f[i]) * f[j];
                        MA_PIN(acc);
                    }
                });
            }
            if constexpr (C == NEG) {                                  // Q - v_0
                acc += (uint64_t)(h == 0 ? (HM + 1u) - u[0] : HM - u[1]);
            } else if constexpr (C > NEG) {                            // s = mask - v_{C-NEG} + v_{C-N}
                uint32_t sh = HM;
                if constexpr (C - NEG <= N) sh -= u[2 * (C - NEG) + h];
                if constexpr (C >= N) sh += u[2 * (C - N) + h];
                acc += (uint64_t)sh;
            }
            u[k] = (uint32_t)acc & HM;
            cy = acc >> H;
        });
        static_for<0, 2 * (N - 1)>([&](auto I) { c[I] = u[2 * (N + 1) + I]; });
        const uint64_t top = cy + ((uint64_t)u[2 * N] | ((uint64_t)u[2 * N + 1] << H)) - 1u;
        c[NL - 2] = (uint32_t)top & HM;
        c[NL - 1] = (uint32_t)(top >> H);
    }
    static MA_DEV void modmul(const uint32_t* a, const uint32_t* b, uint32_t* c) { product<false>(a, b, c); }
    static MA_DEV void modsqr(const uint32_t* a, uint32_t* c) { product<true>(a, a, c); }

    // a * (small positive int) for the trinomial shape (monty.py:1386-1399; Field<P>::monty_modmli, TRIN branch): digits of
    // a * b, the carry out of the top limb re-enters at limbs 0 and TRIN, unmasked.  The running integer is the same in radix
    // 2^28, so two half digits are one of the reference's digits.
    static MA_DEV void modmli(const uint32_t* a, int b, uint32_t* c) {
        static_assert(P::TRIN == 4 && P::XCESS == 0, "2^448 - 2^224 - 1");
        const uint32_t bw = (uint32_t)b;
        uint64_t t = 0;
        uint32_t r[NL];
        static_for<0, NL>([&](auto I) {
            t += (uint64_t)a[I] * bw;
            MA_PIN(t);
            r[I] = (uint32_t)t & HM;
            t >>= H;
        });
        const uint64_t s = t;
        const uint64_t x0 = (uint64_t)r[0] + s, x8 = (uint64_t)r[2 * P::TRIN] + s;
        r[0] = (uint32_t)x0 & HM;
        r[1] += (uint32_t)(x0 >> H);
        r[2 * P::TRIN] = (uint32_t)x8 & HM;
        r[2 * P::TRIN + 1] += (uint32_t)(x8 >> H);
        modcpy(r, c);
    }
};

}  // namespace ma
