// modarith_amd/csrc/ed26l_k.h -- the ED25519 instance of the ladder-form kernel pipeline (csrc/edlad_k.h) on csrc/ed26l.h
#pragma once
#include "edlad_k.h"
#include "ed26l.h"
#include "generated/curve_ED25519.h"

namespace ma {

struct LadT25519 {
    using F = Fe26;
    using P = P_X25519;
    using Lad = Ed26Lad<C_ED25519>;
    static constexpr int NL = 10, NW = 4, NIN = 5;
    static MA_DEV uint32_t prep(const spint* X, const spint* Y, const spint* Z, uint32_t* D, uint32_t* nu, uint32_t* nw) { return Lad::prep(X, Y, Z, D, nu, nw); }
};
using Ed26lWs = EdLadWs<LadT25519>;
inline size_t ed26l_workspace_bytes(size_t n) { return Ed26lWs::bytes(n); }

}  // namespace ma
