// tools/wn26_host.hip -- the per-lane functions of the fused P-256 / secp256k1 kernels (csrc/wn26.h, wj26.h, glv26.h on csrc/fm26.h, fk26.h) compiled for
// the HOST into a small shared library, so that tests/test_host_arith.py can run them against the CPU oracle through
// ctypes (the secp256k1 oracle is bound to its parameter block at run time by tests/oracle_binding.py, which a C main
// cannot do).  Test tooling, not product code.
//   hipcc -O2 -std=c++17 -shared -fPIC --offload-host-only tools/wn26_host.hip -o /tmp/libwn26_host.so
#define MA_DEV __host__ __device__ inline
#include "../modarith_amd/csrc/wn26.h"
#include "../modarith_amd/csrc/generated/comb_NIST256.h"
#include "../modarith_amd/csrc/generated/comb_SECP256K1.h"

extern "C" void secp256k1_mul_get_host(const uint64_t* ew, const uint64_t* X, const uint64_t* Y, const uint64_t* Z, uint64_t* xw, uint64_t* yw) {
    uint64_t tab[ma::WN26_TABLE_WORDS];
    ma::wn26_mul_get_one<ma::CvSecp256k1>(ew, X, Y, Z, tab, 1, xw, yw);
}
extern "C" void secp256k1_mul2_get_host(const uint64_t* ew, const uint64_t* PX, const uint64_t* PY, const uint64_t* PZ,
                                        const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ, uint64_t* xw, uint64_t* yw) {
    uint64_t tab[ma::WN26_TABLE_WORDS];
    ma::wn26_mul2_get_one<ma::CvSecp256k1>(ew, PX, PY, PZ, fw, QX, QY, QZ, tab, 1, xw, yw);
}

// field-level entry points for the limb-bound tests (tests/test_host_arith.py): one product (MODE as in fm26.h / fk26.h)
// of lazy signed limbs, exported canonically.  which = 0: Fm26 (P-256, result = value / R'^2 mod p), 1: Fk26 (secp256k1)
extern "C" void wn26_field_product(int which, int mode, const int32_t* f, const int32_t* g, const int32_t* u, const int32_t* v, uint64_t* w) {
    int32_t r[10];
    if (which == 0) {
        if (mode == 0) ma::Fm26::mul(f, g, r);
        else if (mode == 1) ma::Fm26::sqr(f, r);
        else ma::Fm26::mul2(f, g, u, v, r);
        ma::Fm26::to_words(r, w);
    } else {
        if (mode == 0) ma::Fk26::mul(f, g, r);
        else if (mode == 1) ma::Fk26::sqr(f, r);
        else ma::Fk26::mul2(f, g, u, v, r);
        ma::Fk26::to_words(r, w);
    }
}

// fused generator multiplication (wn26_mulgen_get_one) with the fixed-base tables as host arrays.  which = 0: P-256, 1: secp256k1
static const int32_t comb_nist256_host[] = { COMB_NIST256_VALUES };
static const int32_t comb_secp256k1_host[] = { COMB_SECP256K1_VALUES };
struct HostCombNist256 { static constexpr int W = COMB_NIST256_W, NW = COMB_NIST256_WINDOWS; static int32_t get(int idx) { return comb_nist256_host[idx]; } };
struct HostCombSecp256k1 { static constexpr int W = COMB_SECP256K1_W, NW = COMB_SECP256K1_WINDOWS; static int32_t get(int idx) { return comb_secp256k1_host[idx]; } };
extern "C" void wn26_mulgen_get_host(int which, const uint64_t* ew, uint64_t* xw, uint64_t* yw) {
    if (which == 0) ma::wn26_mulgen_get_one<ma::CvNist256, HostCombNist256>(ew, xw, yw);
    else ma::wn26_mulgen_get_one<ma::CvSecp256k1, HostCombSecp256k1>(ew, xw, yw);
}
// G scalars with one shared inversion (the kernels' form); e, x, y: G x 4 words
template <class CV, class TAB, int G>
static void many(const uint64_t* e, uint64_t* x, uint64_t* y) {
    uint64_t xw[G][4], yw[G][4];
    ma::wn26_mulgen_get_many<CV, TAB, G>([&](int g, uint64_t* ew) { for (int k = 0; k < 4; k++) ew[k] = e[4 * g + k]; }, xw, yw);
    for (int g = 0; g < G; g++) for (int k = 0; k < 4; k++) { x[4 * g + k] = xw[g][k]; y[4 * g + k] = yw[g][k]; }
}
extern "C" void wn26_mulgen_get_many_host(int which, int G, const uint64_t* e, uint64_t* x, uint64_t* y) {
    if (which == 0) { if (G == 2) many<ma::CvNist256, HostCombNist256, 2>(e, x, y); else many<ma::CvNist256, HostCombNist256, 4>(e, x, y); }
    else { if (G == 2) many<ma::CvSecp256k1, HostCombSecp256k1, 2>(e, x, y); else many<ma::CvSecp256k1, HostCombSecp256k1, 4>(e, x, y); }
}
extern "C" void wn26_mulgen2_get_host(int which, const uint64_t* ew, const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ,
                                      uint64_t* xw, uint64_t* yw) {
    uint64_t tab[ma::WN26_TABLE_WORDS];
    if (which == 0) ma::wn26_mulgen2_get_one<ma::CvNist256, HostCombNist256>(ew, fw, QX, QY, QZ, tab, 1, xw, yw);
    else ma::wn26_mulgen2_get_one<ma::CvSecp256k1, HostCombSecp256k1>(ew, fw, QX, QY, QZ, tab, 1, xw, yw);
}

// round 5: the secp256k1 endomorphism (csrc/glv26.h).  split: |k1|, |k2| (three words each) and the signs (bit 0: k1 < 0, bit 1: k2 < 0)
#include "../modarith_amd/csrc/glv26.h"
extern "C" int secp256k1_glv_split_host(const uint64_t* ew, uint64_t* k1, uint64_t* k2) {
    bool n1, n2;
    ma::GlvSecp256k1::split(ew, k1, n1, k2, n2);
    return (n1 ? 1 : 0) | (n2 ? 2 : 0);
}
extern "C" void secp256k1_glv_mul_get_host(const uint64_t* ew, const uint64_t* X, const uint64_t* Y, const uint64_t* Z, uint64_t* xw, uint64_t* yw) {
    uint64_t tab[ma::WN26_TABLE_WORDS];
    ma::GlvRegs dig;
    dig.init(ew);
    ma::secp256k1_glv_mul_get_dig(dig, X, Y, Z, ma::WnTabStrided{tab, 1}, xw, yw);
}
extern "C" void secp256k1_glv_mulgen2_get_host(const uint64_t* ew, const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ,
                                               uint64_t* xw, uint64_t* yw) {
    uint64_t tab[ma::WN26_TABLE_WORDS];
    ma::GlvRegs dig;
    dig.init(fw);
    ma::secp256k1_glv_mulgen2_get_dig<HostCombSecp256k1>(ew, dig, QX, QY, QZ, ma::WnTabStrided{tab, 1}, xw, yw);
}
extern "C" void secp256k1_glv_mul2_get_host(const uint64_t* ew, const uint64_t* PX, const uint64_t* PY, const uint64_t* PZ,
                                            const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ, uint64_t* xw, uint64_t* yw) {
    uint64_t tab[ma::GLV2_TABLE_WORDS];
    ma::GlvRegs de, df;
    de.init(ew);
    df.init(fw);
    ma::secp256k1_glv_mul2_get_dig(de, PX, PY, PZ, df, QX, QY, QZ, ma::WnTabStrided{tab, 1}, xw, yw);
}

// round 5: P-256 in Jacobian coordinates on affine window tables (csrc/wj26.h, wn_affine.h): the kernels' pipeline for ONE record -- the table
// of the point (two for e P + f Q), the normalisation, the window loop -- then wn26.h's export
#include "../modarith_amd/csrc/wj26.h"
static auto nist_cp(const uint64_t* a, const uint64_t* b, const uint64_t* c) {
    return [=](uint64_t* x, uint64_t* y, uint64_t* z) { for (int i = 0; i < 5; i++) { x[i] = a[i]; y[i] = b[i]; z[i] = c[i]; } };
}
static void nist_kP(const uint64_t* ew, const uint64_t* X, const uint64_t* Y, const uint64_t* Z, ma::Wj26::Pt& R) {
    static uint64_t buf[(8 * 15 + 8 * 5) + 1];
    ma::WnAffWs ws(buf, 1);
    ma::Wj26::table_of(nist_cp(X, Y, Z), ws, 0);
    ma::wn_table_affine_lane<ma::Fm26, true>(ws, 1, 1, 0);
    uint64_t k[4];
    ma::Wj26::reduce_scalar(ew, k);
    ma::WnRegs<4, 260> dig;
    dig.init(k);
    ma::Wj26::mul_acc_aff(dig, ws, 0, R);
}
// the affine table (8 entries x 10 words: x, y) of record 0 when it shares its inversion with record 1 (m = 2 records in one lane's
// column: L = 1, R = 2) -- tests/test_host_arith.py puts a point off the curve next to a good one
extern "C" void nist256_affine_table_pair_host(const uint64_t* X0, const uint64_t* Y0, const uint64_t* Z0, const uint64_t* X1, const uint64_t* Y1,
                                               const uint64_t* Z1, uint64_t* table0, uint32_t* flags) {
    static uint64_t buf[2 * (8 * 15 + 8 * 5) + 2];
    ma::WnAffWs ws(buf, 2);
    ma::Wj26::table_of(nist_cp(X0, Y0, Z0), ws, 0);
    ma::Wj26::table_of(nist_cp(X1, Y1, Z1), ws, 1);
    ma::wn_table_affine_lane<ma::Fm26, true>(ws, 1, 2, 0);
    for (int e = 0; e < 8; e++)
        for (int k = 0; k < 10; k++) table0[e * 10 + k] = ws.T[(size_t)(e * 15 + k) * 2 + 0];
    flags[0] = ws.flag[0];
    flags[1] = ws.flag[1];
}
extern "C" void nist256_jac_mul_get_host(const uint64_t* ew, const uint64_t* X, const uint64_t* Y, const uint64_t* Z, uint64_t* xw, uint64_t* yw) {
    ma::Wj26::Pt R;
    nist_kP(ew, X, Y, Z, R);
    ma::Wn26<ma::CvNist256>::affine_words(R, xw, yw);
}
extern "C" void nist256_jac_mulgen2_get_host(const uint64_t* ew, const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ,
                                             uint64_t* xw, uint64_t* yw) {
    ma::Wj26::Pt R;
    nist_kP(fw, QX, QY, QZ, R);
    ma::wn26_mulgen_acc<ma::CvNist256, HostCombNist256, false>(ew, R);
    ma::Wn26<ma::CvNist256>::affine_words(R, xw, yw);
}
extern "C" void nist256_jac_mul2_get_host(const uint64_t* ew, const uint64_t* PX, const uint64_t* PY, const uint64_t* PZ,
                                          const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ, uint64_t* xw, uint64_t* yw) {
    static uint64_t buf[(16 * 15 + 16 * 5) + 1];
    ma::WnAffWs ws(buf, 1, 16);
    ma::Wj26::table_of(nist_cp(PX, PY, PZ), ws, 0, 0);
    ma::Wj26::table_of(nist_cp(QX, QY, QZ), ws, 0, 8);
    ma::wn_table_affine_lane<ma::Fm26, true>(ws, 1, 1, 0);
    ma::WnRegs<4, 260> de, df;
    de.init(ew);
    df.init(fw);
    ma::Wj26::Pt R;
    ma::Wj26::mul2_acc_aff(de, df, ws, 0, R);
    ma::Wn26<ma::CvNist256>::affine_words(R, xw, yw);
}
extern "C" void nist256_jac_mulgen_get_host(const uint64_t* ew, uint64_t* xw, uint64_t* yw) {
    uint64_t k[4];
    ma::Wj26::reduce_scalar(ew, k);
    ma::Wj26::Pt R;
    ma::Wj26::mulgen_acc<HostCombNist256>(k, R);
    ma::Wn26<ma::CvNist256>::affine_words(R, xw, yw);
}
