// modarith_amd/csrc/ed28l_k.h -- the ED448 instance of the ladder-form kernel pipeline (csrc/edlad_k.h) on csrc/ed28l.h
#pragma once
#include "edlad_k.h"
#include "ed28l.h"
#include "generated/params_X448.h"

namespace ma {

struct LadT448 {
    using F = Fe28;
    using P = P_X448;
    using Lad = Ed28Lad;
    static constexpr int NL = 16, NW = 7, NIN = 8;
    static MA_DEV uint32_t prep(const spint* X, const spint* Y, const spint* Z, uint32_t* D, uint32_t* nu, uint32_t* nw) { return Lad::prep(X, Y, Z, D, nu, nw); }
};
using Ed28lWs = EdLadWs<LadT448>;
inline size_t ed28l_workspace_bytes(size_t n) { return Ed28lWs::bytes(n); }

}  // namespace ma
