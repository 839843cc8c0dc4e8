// modarith_amd/csrc/fh51.h -- the 2^255-19 field of field.c (5 x 51-bit limbs, pseudo.py) with every element RESIDENT in
// half-limb form: ten 32-bit words h[0..9], limb k = h[2k] + 2^26 h[2k+1], h[2k] < 2^26 always; h[2k+1] carries whatever the
// limb holds above bit 26 (25 bits for a masked digit, more where the reference leaves a limb unmasked).  It is the SAME
// element with the SAME limbs as Field<P_X25519> holds -- from_limbs() / to_limbs() are exact both ways -- and every function
// returns exactly the limbs the reference's function returns (pseudo.py:286-348 modadd/modsub/modneg generic forms, 616-702
// modmul/modsqr with the second pass 557-611) for field elements as the API defines them (SURVEY 8c caveat 2): limbs inside the
// contract of the FAST path (< 2^53), VALUE below 2p for the sums (for larger values the reference's modsub / modneg leave a
// negative top limb -- 64 bits wide there, 32 here -- which no function of the API accepts); tools/fe_host_check.hip run_fh51.
//
// Why: the scalar multiplications of the curve layer (csrc/curve.h) are chains of ~2 600 field multiplications with ~2 300
// additions between them.  Field<P>::pm_modmul_half already multiplies on half limbs, but cuts each 64-bit limb into halves
// on the way in (and / alignbit / and per limb and operand) and glues the half digits back on the way out (64-bit shift + or
// per limb): ~130 of the ~290 non-multiplier cycles of a product.  Held in half form between the operations those conversions
// disappear; additions cost the same (their carry chains pass 9 narrow boundaries instead of 4 wide ones, in 32-bit
// instructions at half the issue cost of the 64-bit ones).  Used by k_ed_mul / k_ed_mul2 / k_ed_mul2x of the curves over this
// field; HBM and the window tables keep the reference's limbs (converted at load / store, a table word packs one limb's halves).
#pragma once
#include "field.h"

namespace ma {

template <class P>
struct FieldH51 {
    static_assert(!P::MONTGOMERY && P::EPM && !P::OVERFLOW && P::FRED && !P::CARRY_ON && P::XCESS == 0 && P::RADIX == 51 && P::N == 5 && P::MM == 19 && P::M == 19,
                  "FieldH51 is the half-limb resident form of the 5 x 51-bit field 2^255-19");
    using limb_t = uint32_t;
    using L = Field<P, true>;                       // the limb-form functions (same element, same limbs)
    static constexpr int N = 5;                     // limbs of the element (HBM form)
    static constexpr int NL = 10;                   // resident words
    static constexpr uint32_t M26 = (1u << 26) - 1u, M25 = (1u << 25) - 1u;
    static constexpr int hbits(int i) { return (i & 1) ? 25 : 26; }
    static constexpr uint32_t hmask(int i) { return (i & 1) ? M25 : M26; }

    // Exact for limbs below 2^58 (h[2k+1] = a_k >> 26 fits 32 bits), i.e. far beyond the limb budget 2^(Radix+2) = 2^53 of the curve
    // layer; a limb of 2^58 or more loses its top bits here.  It is NOT normalised (as fh56.h does for the Montgomery form): the
    // reference's pseudo-Mersenne product folds columns, so its output limbs depend on the limbs it is given, not only on the integer
    // -- a representative with bits 51, 52 set inside the budget must go in as it is.  include/modarith_amd.h states the contract.
    static MA_DEV void from_limbs(const spint* a, uint32_t* h) {
        static_for<0, N>([&](auto K) {
            h[2 * K] = (uint32_t)a[K] & M26;
            h[2 * K + 1] = (uint32_t)(a[K] >> 26);
        });
    }
    static MA_DEV void to_limbs(const uint32_t* h, spint* a) {
        static_for<0, N>([&](auto K) { a[K] = (spint)h[2 * K] + ((spint)h[2 * K + 1] << 26); });
    }
    // one 64-bit word per limb for the window tables: the two halves side by side (no shifts on either side)
    static MA_DEV spint pack(const uint32_t* h, int k) { return (spint)h[2 * k] | ((spint)h[2 * k + 1] << 32); }
    static MA_DEV void unpack(spint w, uint32_t* h, int k) { h[2 * k] = (uint32_t)w; h[2 * k + 1] = (uint32_t)(w >> 32); }

    static MA_DEV void modcpy(const uint32_t* a, uint32_t* c) { static_for<0, NL>([&](auto I) { c[I] = a[I]; }); }
    static MA_DEV void modzer(uint32_t* a) { static_for<0, NL>([&](auto I) { a[I] = 0; }); }
    static MA_DEV void modone(uint32_t* a) { a[0] = 1; static_for<1, NL>([&](auto I) { a[I] = 0; }); }
    static MA_DEV void modcmv(int b, const uint32_t* g, uint32_t* f) {
        const bool take = (b & 1) != 0;
        static_for<0, NL>([&](auto I) {
            const uint32_t x = g[I], y = f[I];
            f[I] = take ? x : y;
        });
    }

    // ---------------------------------------------------------------- add / sub / neg (generic=True forms)
    // The reference: n = a (+-) b limb-wise, (modadd: n -= 2p,) prop, n += 2p under the sign mask of the top limb, prop
    // (pseudo.py:286-348).  -2p = +38 on limb 0, -2^52 on limb 4 (= -2^26 on h[9]).  Both props run here as 32-bit chains over
    // the nine half boundaries; the first one is fused with the limb-wise sum (v_add3_u32), carries are arithmetic shifts as in
    // prop (pseudo.py:223-251), and the top word h[9] stays unmasked as the top limb does.  Same integer at every step, hence
    // the same digits.  Inputs: h[i] < 2^30 (any element of this form inside the contract is far below).
    // RIPPLE = false: the "_u" forms (see Field<P>::modadd_u): the closing chain is left out, n[0] and the top word carry the +2p
    template <bool RIPPLE = true, class First>
    static MA_DEV void chains(First first, uint32_t* n) {
        // chain 1: n[i] = first(i) + carry
        int32_t c = 0;
        static_for<0, NL - 1>([&](auto I) {
            const int32_t x = first(I) + c;
            n[I] = (uint32_t)x & hmask(I);
            c = x >> hbits(I);
        });
        int32_t top = first(std::integral_constant<int, NL - 1>{}) + c;
        const int32_t m = top >> 31;                         // all ones if the value is negative
        if constexpr (!RIPPLE) {
            n[0] = (uint32_t)((int32_t)n[0] - (38 & m));     // (may go negative: the consumer's first chain takes it as a signed word)
            n[NL - 1] = (uint32_t)(top + ((1 << 26) & m));
            return;
        }
        // + 2p under the mask, chain 2
        int32_t y = (int32_t)n[0] - (38 & m);
        n[0] = (uint32_t)y & M26;
        c = y >> 26;
        static_for<1, NL - 1>([&](auto I) {
            const int32_t x = (int32_t)n[I] + c;
            n[I] = (uint32_t)x & hmask(I);
            c = x >> hbits(I);
        });
        n[NL - 1] = (uint32_t)(top + ((1 << 26) & m) + c);
    }
    static MA_DEV void modadd(const uint32_t* a, const uint32_t* b, uint32_t* n) {
        uint32_t r[NL];
        chains([&](auto I) -> int32_t {
            constexpr int i = I;
            return (int32_t)(a[i] + b[i]) + (i == 0 ? 38 : 0) - (i == NL - 1 ? (1 << 26) : 0);
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
            return (int32_t)(a[i] + b[i]) + (i == 0 ? 38 : 0) - (i == NL - 1 ? (1 << 26) : 0);
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
    // second pass (pseudo.py:557-611; FRED form, no carry-on): ut = 19 * t; limb 0 takes ut's low 51 bits, limb 1 the carry
    // (s >> 51) + (ut >> 51) and stays unmasked.  t = the value above bit 255, d = the ten half digits below.
    static MA_DEV void second_pass(uint64_t t, const uint32_t* d, uint32_t* c) {
        const uint64_t ut = t * (uint64_t)P::M;
        const uint32_t ulo = (uint32_t)ut & M26;
        const uint32_t umid = (uint32_t)(ut >> 26) & M25;
        const uint32_t uhi = (uint32_t)(ut >> 51);
        const uint32_t s0 = d[0] + ulo;
        const uint32_t s1 = d[1] + umid + (s0 >> 26);
        const uint32_t s2 = d[2] + uhi + (s1 >> 25);
        c[0] = s0 & M26;
        c[1] = s1 & M25;
        c[2] = s2 & M26;
        c[3] = d[3] + (s2 >> 26);
        static_for<4, NL>([&](auto I) { c[I] = d[I]; });
    }
    // c = a * b: the column scheme of Field<P>::pm_modmul_half (see the bounds and the odd-odd note there), operands and result
    // in half form
    static MA_DEV void modmul(const uint32_t* f, const uint32_t* g, uint32_t* c) {
        constexpr int M = NL;
        uint32_t g19[M], f2[M], t[M];
        static_for<1, M>([&](auto J) { g19[J] = (uint32_t)P::MM * g[J]; });
        static_for<0, N>([&](auto K) { f2[2 * K + 1] = 2u * f[2 * K + 1]; });
        uint64_t cy = 0;
        static_for<0, M>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = cy;
            static_for<0, M>([&](auto II) {
                constexpr int i = II;
                constexpr int j = (k - i + M) % M;
                constexpr bool wrp = (i + j) >= M;
                constexpr bool dbl = (i & 1) && (j & 1);
                if constexpr (!(dbl && i + j == M)) {             // (those go on top of the last carry, below)
                    const uint32_t x = dbl ? f2[i] : f[i];
                    const uint32_t y = wrp ? g19[j] : g[j];
                    acc += (uint64_t)x * y;
                    MA_PIN(acc);
                }
            });
            t[k] = (uint32_t)acc & hmask(k);
            cy = acc >> hbits(k);
        });
        static_for<0, N>([&](auto K) {
            constexpr int i = 2 * K + 1;
            cy += (uint64_t)f2[i] * g[M - i];
            MA_PIN(cy);
        });
        second_pass(cy, t, c);
    }
    static MA_DEV void modsqr(const uint32_t* f, uint32_t* c) {
        constexpr int M = NL;
        uint32_t f2[M], f4[M], f19[M], t[M];
        static_for<0, M>([&](auto I) { f2[I] = 2u * f[I]; });
        static_for<0, N>([&](auto K) { f4[2 * K + 1] = 4u * f[2 * K + 1]; });
        static_for<1, M>([&](auto J) { f19[J] = (uint32_t)P::MM * f[J]; });
        uint64_t cy = 0;
        static_for<0, M>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = cy;
            static_for<0, M>([&](auto II) {
                constexpr int i = II;
                constexpr int j = (k - i + M) % M;
                if constexpr (i <= j) {
                    constexpr bool wrp = (i + j) >= M;
                    constexpr bool odd2 = (i & 1) && (j & 1);
                    uint32_t x, y;
                    if constexpr (i == j) {                       // f_i^2, times 2 if odd, times MM if wrapped
                        x = odd2 ? f2[i] : f[i];
                        y = wrp ? f19[j] : f[j];
                    } else {                                      // 2 f_i f_j, times 2 if both odd, times MM if wrapped
                        x = odd2 ? f4[i] : f2[i];
                        y = wrp ? f19[j] : f[j];
                    }
                    if constexpr (!(odd2 && i + j == M)) {
                        acc += (uint64_t)x * y;
                        MA_PIN(acc);
                    }
                }
            });
            t[k] = (uint32_t)acc & hmask(k);
            cy = acc >> hbits(k);
        });
        static_for<0, N>([&](auto K) {                            // odd i + j = 2N: on top of the last carry
            constexpr int i = 2 * K + 1, j = M - i;
            if constexpr (i < j) { cy += (uint64_t)f4[i] * f[j]; MA_PIN(cy); }
            else if constexpr (i == j) { cy += (uint64_t)f2[i] * f[j]; MA_PIN(cy); }
        });
        second_pass(cy, t, c);
    }
    // a * (small positive int), pseudo.py:705-728, through the limb form (the curves over this field that use it are not hot)
    static MA_DEV void modmli(const uint32_t* a, int b, uint32_t* c) {
        spint x[N];
        to_limbs(a, x);
        L::modmli(x, b, x);
        from_limbs(x, c);
    }
};

}  // namespace ma
