"""Fused chains: a sequence of field.c calls per element compiled into ONE streaming kernel.

A C caller of the reference writes `modadd(x,y,t); modsub(x,y,w); modmul(t,w,s); modsqr(s,s); modinv(s,NULL,z);`
(the generators' acceptance chain, pseudo.py:1783-1796; the bodies of rfc7748.c:190-223, edwards.c:73-145) and the C
compiler keeps every intermediate in registers -- field.c's functions are `static inline`.  Batched over HBM-resident
arrays, the same five calls are five kernels and 520 bytes of traffic per element where 120 are needed (two arrays in,
one out): on a memory-bound engine the round trips are the whole cost.  This module gives the batched caller what the C
compiler gives the scalar one: the chain is written once with the reference's function names,

    ch = Chain("X25519", "accept")             # any built-in prime or generated tag
    x, y = ch.input(), ch.input()
    s = ch.modsqr(ch.modmul(ch.modadd(x, y), ch.modsub(x, y)))
    ch.output(ch.modinv(s))
    f = ch.build()                              # one hipcc unit (seconds), cached
    z, = f(xa, ya)                              # device batches, flat or tiled

and hipcc instantiates the same `Field<P>` functions the library's kernels are made of, back to back, on registers:
one load per input array, one store per output array, no LDS, no interpreter.  Limbs are those of the call-by-call
sequence, bit for bit (the same functions on the same limbs; `modinv` returns the library's normalised form).  The
product policy is the library's wave vote (csrc/kernels.h OpMulAuto), taken once on the inputs: inside the limb contract
the whole chain runs the split products -- every intermediate is a field-function output and stays inside it -- otherwise
the exact ones.  C-ABI of the built plug-in (modarith_amd/plugins/libmodarith_amd_chain_<name>_<TAG>.so):

    int chain_<name>_<TAG>_batch(const void *const *in, void *const *out, size_t n, size_t ld, void *stream);
    int chain_<name>_<TAG>_aos(const void *const *in, void *const *out, size_t n, void *stream);     /* element-major x[n][Nlimbs] */

(`in`: the element batches, then one int32 array per selector of modcmv / modcsw.)  Operations: modmul modsqr modadd modsub modneg
modmli nres redc modcpy modinv modpro modsqrt modnsqr modhaf modcmv modcsw and the generic=False forms modadd_lazy modsub_lazy
modneg_lazy (one level of laziness between reductions, which is how rfc7748.c uses them).

Like the generator mode (modarith_amd/generate.py) this needs hipcc where the chain is built and has no CPU path.
"""
from __future__ import annotations

import ctypes
import hashlib
import json
import os
import re
import subprocess
from typing import List, Optional, Sequence

from . import _lib, emit
from . import generate as _gen

HERE = os.path.dirname(os.path.abspath(__file__))
_NAME_RE = re.compile(r"^[A-Za-z][A-Za-z0-9_]*$")
# op -> (Field<P> function, operand count, takes an int immediate)
_OPS = {"modmul": 2, "modadd": 2, "modsub": 2, "modsqr": 1, "modneg": 1, "nres": 1, "redc": 1, "modcpy": 1, "modinv": 1, "modmli": 1,
        "modadd_lazy": 2, "modsub_lazy": 2, "modneg_lazy": 1, "modnsqr": 1, "modpro": 1, "modsqrt": 1, "modhaf": 1, "modcmv": 2, "modcsw": 2}
_HEAVY = ("modinv", "modpro", "modsqrt")          # long exponentiation chains: one element per lane
_LAZY = ("modadd_lazy", "modsub_lazy", "modneg_lazy")
MAX_OPS = 256


class Val:
    """one element-valued intermediate of a chain (an SSA value: written once)"""
    def __init__(self, chain: "Chain", idx: int):
        self.chain, self.idx = chain, idx


class Sel:
    """a per-element int selector input of a chain"""
    def __init__(self, chain: "Chain", idx: int):
        self.chain, self.idx = chain, idx


class Chain:
    def __init__(self, prime: str, name: str):
        if not _NAME_RE.match(name):
            raise ValueError("chain name %r cannot be part of a C identifier" % (name,))
        if prime in _lib.PRIMES:
            from .params import derive
            self.params = derive(prime)
            self.builtin = True
        elif os.path.exists(_gen.plugin_path(prime)):
            self.params = _gen.params_of_plugin(prime)
            self.builtin = False
        else:
            raise ValueError("prime %r is neither built in nor generated (python -m modarith_amd.generate 64 <prime>)" % (prime,))
        self.prime, self.name = prime, name
        self.nin = 0
        self.nsel = 0                       # per-element int selectors (modcmv / modcsw): int32 arrays after the element inputs
        self.lazy = set()                   # values produced by a generic=False operation
        self.ops: List[tuple] = []          # (op, dst, a, b, imm)
        self.outs: List[int] = []
        self.nvals = 0

    # ------------------------------------------------------------------ building
    def _new(self) -> Val:
        v = Val(self, self.nvals)
        self.nvals += 1
        return v

    def input(self) -> Val:
        if self.ops:
            raise ValueError("declare every input before the first operation")
        self.nin += 1
        return self._new()

    def inputs(self, k: int) -> List[Val]:
        return [self.input() for _ in range(k)]

    def _own(self, *vs: Val):
        for v in vs:
            if not isinstance(v, Val) or v.chain is not self:
                raise ValueError("operands must be values of this chain")

    def _op(self, op: str, a: Val, b: Optional[Val] = None, imm: int = 0) -> Val:
        self._own(a, *([b] if b is not None else []))
        if len(self.ops) >= MAX_OPS:
            raise ValueError("chains are limited to %d operations" % MAX_OPS)
        d = self._new()
        self.ops.append((op, d.idx, a.idx, b.idx if b is not None else -1, int(imm)))
        return d

    def modmul(self, a, b): return self._op("modmul", a, b)
    def modadd(self, a, b): return self._op("modadd", a, b)
    def modsub(self, a, b): return self._op("modsub", a, b)
    def modsqr(self, a): return self._op("modsqr", a)
    def modneg(self, a): return self._op("modneg", a)
    def nres(self, a): return self._op("nres", a)
    def redc(self, a): return self._op("redc", a)
    def modcpy(self, a): return self._op("modcpy", a)
    def modinv(self, a): return self._op("modinv", a)

    def modmli(self, a, k: int):
        if not -(1 << 31) <= int(k) < (1 << 31):
            raise ValueError("modmli takes a C int")
        return self._op("modmli", a, imm=k)

    # generic=False forms (pseudo.py:294-302, 315-324, 337-346: what rfc7748.c:20 asks for).  Their results carry up to two extra bits;
    # one level of them keeps the limb contract the split products rely on, a lazy sum of lazy sums need not -- refused here
    def _lazy(self, op, a, b=None):
        for v in (a, b):
            if v is not None and v.idx in self.lazy:
                raise ValueError("%s of a value that is itself a generic=False result: reduce in between (modadd / modsub / modmul ...)" % op)
        d = self._op(op, a, b)
        self.lazy.add(d.idx)
        return d

    def modadd_lazy(self, a, b): return self._lazy("modadd_lazy", a, b)
    def modsub_lazy(self, a, b): return self._lazy("modsub_lazy", a, b)
    def modneg_lazy(self, a): return self._lazy("modneg_lazy", a)
    def modpro(self, a): return self._op("modpro", a)
    def modsqrt(self, a): return self._op("modsqrt", a)
    def modhaf(self, a): return self._op("modhaf", a)

    def modnsqr(self, a, k: int):
        if not 0 <= int(k) <= 100000:
            raise ValueError("modnsqr count out of range")
        return self._op("modnsqr", a, imm=k)

    def selector(self) -> "Sel":
        """a per-element int input (0 / 1), as the `int b` of modcmv / modcsw: an int32 array, passed after the element inputs"""
        if self.ops:
            raise ValueError("declare every input before the first operation")
        self.nsel += 1
        return Sel(self, self.nsel - 1)

    def _sel(self, d):
        if not isinstance(d, Sel) or d.chain is not self:
            raise ValueError("the selector must come from this chain's selector()")
        return d.idx

    def modcmv(self, d, g, f):
        """f' = d ? g : f  (constant time: lane-predicated selects, pseudo.py:1017-1048)"""
        return self._op("modcmv", g, f, imm=self._sel(d))

    def modcsw(self, d, g, f):
        """(g', f') = d ? (f, g) : (g, f)  (pseudo.py:979-1014)"""
        k = self._sel(d)
        self._own(g, f)
        g2, f2 = self._new(), self._new()
        self.ops.append(("modcsw", g2.idx, g.idx, f.idx, k, f2.idx))
        return g2, f2

    def output(self, v: Val) -> None:
        self._own(v)
        self.outs.append(v.idx)

    # ------------------------------------------------------------------ emission
    def default_ept(self) -> int:
        return 1 if any(o[0] in _HEAVY for o in self.ops) else 2

    @property
    def symbol(self) -> str:
        return "chain_%s_%s_batch" % (self.name, self.prime)

    def traffic_bytes(self) -> int:
        """HBM bytes per element of the fused kernel; the call-by-call sequence moves sum(8 N (operands + 1)) instead"""
        return 8 * self.params.nlimbs * (self.nin + len(self.outs)) + 4 * self.nsel

    def unfused_traffic_bytes(self) -> int:
        return sum(8 * self.params.nlimbs * (_OPS[o[0]] + (2 if o[0] == "modcsw" else 1)) + (4 if o[0] in ("modcmv", "modcsw") else 0) for o in self.ops)

    def source(self, ept: Optional[int] = None, policy: str = "vote", waves: int = 0) -> str:
        """ept: elements per lane on aligned batches (2 = 16-byte accesses, 1 = 8-byte); None = the measured default"""
        if not self.nin or not self.outs:
            raise ValueError("a chain needs at least one input and one output")
        P, nv = self.prime, self.nvals
        L = ["// GENERATED by modarith_amd/fuse.py -- do not edit.  Chain %r over %s: %d inputs, %d operations, %d outputs." % (self.name, P, self.nin, len(self.ops), len(self.outs)),
             '#include "params_%s.h"' % P, '#include "modarith_amd.h"', '#include "capi_common.h"', '#include "kernels.h"', "",
             "namespace {", "using namespace ma;", "using P = ma::P_%s;" % P, "constexpr int NIN = %d, NOUT = %d;" % (self.nin, len(self.outs)),
             "constexpr bool HEAVY = %s;   // one element per lane (8-byte accesses) on every batch: chains with an inversion, as the library's k_unary_heavy" % ("true" if (ept or self.default_ept()) == 1 else "false"),
             "constexpr int NSEL = %d;" % self.nsel,
             "struct Args { const spint* in[NIN]; spint* out[NOUT]; const int* sel[NSEL > 0 ? NSEL : 1]; };", "",
             "// the chain on one element's registers; F = Field<P, FAST>",
             "template <class F> MA_DEV void body(%s) {" % ", ".join([("const spint* v%d" if i < self.nin else "spint* v%d") % i for i in range(nv)] + ["int s%d" % k for k in range(self.nsel)])]
        for o in self.ops:
            op, d, a, b, imm = o[:5]
            if op == "modcsw":
                L.append("    F::modcpy(v%d, v%d); F::modcpy(v%d, v%d); F::modcsw(s%d, v%d, v%d);" % (a, d, b, o[5], imm, d, o[5]))
            elif op == "modcmv":
                L.append("    F::modcpy(v%d, v%d); F::modcmv(s%d, v%d, v%d);" % (b, d, imm, a, d))
            elif op == "modnsqr":
                L.append("    F::modcpy(v%d, v%d); F::modnsqr(v%d, %d);" % (a, d, d, imm))
            elif op == "modhaf":
                L.append("    F::modcpy(v%d, v%d); F::modhaf(v%d);" % (a, d, d))
            elif op == "modsqrt":
                L.append("    F::modsqrt(v%d, nullptr, v%d);" % (a, d))
            elif op == "modinv":
                L.append("    F::modinv(v%d, nullptr, v%d); inv_normalise<F>(v%d);" % (a, d, d))
            elif op == "modmli":
                L.append("    F::modmli(v%d, %d, v%d);" % (a, imm, d))
            elif _OPS[op] == 2:
                L.append("    F::%s(v%d, v%d, v%d);" % (op, a, b, d))
            else:
                L.append("    F::%s(v%d, v%d);" % (op, a, d))
        L += ["}", "",
              "template <int EPT>", "__global__ __launch_bounds__(BLOCK) %svoid k_chain(Args A, size_t nthreads, Ld L) {" % ("__attribute__((amdgpu_waves_per_eu(%d, %d))) " % (waves, waves) if waves else "")]
        votel = ["bool fast = %s;" % ("true" if policy == "fast" else "false"),
                 "if constexpr (P::SPLIT > 0 && %s) {" % ("true" if policy == "vote" else "false"),
                 "    bool ok = true;",
                 "    static_for<0, EPT>([&](auto E) { ok = ok && " + " && ".join("in_split_contract<P>(v%d[E])" % i for i in range(self.nin)) + "; });",
                 "    fast = __all(ok);",
                 "}",
                 "if (fast) {",
                 "    static_for<0, EPT>([&](auto E) { body<Field<P, true>>(%s); });" % ", ".join(["v%d[E]" % i for i in range(nv)] + ["s%d[E]" % k for k in range(self.nsel)]),
                 "} else {",
                 "    static_for<0, EPT>([&](auto E) { body<Field<P, false>>(%s); });" % ", ".join(["v%d[E]" % i for i in range(nv)] + ["s%d[E]" % k for k in range(self.nsel)]),
                 "}"]
        L += ["    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {",
              "        " + " ".join("spint v%d[EPT][P::N];" % i for i in range(nv))]
        L += ["        load_soa<P, EPT>(A.in[%d], L, t, v%d);" % (i, i) for i in range(self.nin)]
        L += ["        int s%d[EPT]; static_for<0, EPT>([&](auto E) { s%d[E] = A.sel[%d][(size_t)EPT * t + E]; });" % (k, k, k) for k in range(self.nsel)]
        L += ["        " + l for l in votel]
        L += ["        store_soa<P, EPT>(A.out[%d], L, t, v%d);" % (k, o) for k, o in enumerate(self.outs)]
        L += ["    }", "}"]
        L += self._aos_kernel(votel)
        L += ["}  // namespace", "",
              'extern "C" int %s(const void* const* in, void* const* out, size_t n, size_t ld, void* stream) {' % self.symbol,
              "    // in[0 .. NIN): element batches; in[NIN .. NIN + NSEL): int32 selector arrays, one entry per element",
              "    if (n == 0) return 0;",
              "    Ld L(ld);",
              "    if (ld < n) {",
              '        if (ld < 128 || (ld & (ld - 1)) != 0) { set_error("%s: a limb stride below n selects the tiled layout and must be a power of two >= 128"); return (int)hipErrorInvalidValue; }' % self.symbol,
              "        L = Ld(ld, (unsigned)__builtin_ctzll((unsigned long long)ld));",
              "    }",
              "    Args A;",
              "    bool al = ld % 2 == 0;",
              "    for (int i = 0; i < NIN; i++) { A.in[i] = (const spint*)in[i]; al = al && aligned16(in[i]); }",
              "    for (int i = 0; i < NOUT; i++) { A.out[i] = (spint*)out[i]; al = al && aligned16(out[i]); }",
              "    for (int i = 0; i < NSEL; i++) A.sel[i] = (const int*)in[NIN + i];",
              "    hipStream_t s = (hipStream_t)stream;",
              "    const bool tiled = L.s != 63;",
              "    if (n >= 2 && al && !HEAVY) {",
              "        size_t nt = n / 2;",
              "        k_chain<2><<<grid_for(nt, BLOCK, tiled), BLOCK, 0, s>>>(A, nt, L);",
              "        if (n & 1) {                                  // the odd element out: one lane on the 8-byte path, at its own address",
              "            const size_t o = L.off<P::N>(n - 1);",
              "            Args B;",
              "            for (int i = 0; i < NIN; i++) B.in[i] = A.in[i] + o;",
              "            for (int i = 0; i < NOUT; i++) B.out[i] = A.out[i] + o;",
              "            for (int i = 0; i < NSEL; i++) B.sel[i] = A.sel[i] + (n - 1);",
              "            k_chain<1><<<1, BLOCK, 0, s>>>(B, 1, Ld(L.ld));",
              "        }",
              "    } else {",
              "        k_chain<1><<<grid_for(n, BLOCK, tiled), BLOCK, 0, s>>>(A, n, L);",
              "    }",
              '    return check_launch("%s");' % self.symbol,
              "}", ""]
        L += self._aos_entry()
        return "\n".join(L)

    @property
    def aos_symbol(self) -> str:
        return "chain_%s_%s_aos" % (self.name, self.prime)

    def _aos_kernel(self, votel: List[str]) -> List[str]:
        """The same chain over ELEMENT-MAJOR arrays (`spint x[n][Nlimbs]`, how the scalar callers of field.c hold their elements): a
        workgroup of 256 lanes takes a chunk of 512 elements; each array's chunk is one linear stretch of 512 N words, moved with
        16-byte-per-lane coalesced accesses and transposed through LDS (pitch N | 1 words: two-way bank conflicts at most) into the
        lanes' registers -- elements 2t, 2t+1 -- and back.  The layout conversions on either side of a chain (aos_to_soa before,
        soa_to_aos after: 160 bytes of HBM traffic per element and array) disappear into the kernel."""
        nv, N = self.nvals, self.params.nlimbs
        if N > 14:                                    # 512 (N | 1) words of LDS per workgroup: 64 KB hold up to 14 limbs (as k_convert_lds)
            return []
        L = ["", "// ---- element-major I/O: x[n][N] in, x[n][N] out, transposed through LDS (the pattern of capi_common.hip k_convert_lds)",
             "constexpr int CH = 512, SP = P::N | 1;",
             "static __device__ __forceinline__ void aos_in(const spint* aos, size_t cnt, spint* sh, spint (*v)[P::N]) {",
             "    const int t = threadIdx.x;",
             "    const size_t total = cnt * (size_t)P::N;",
             "    int e = (2 * t) / P::N, i = (2 * t) % P::N;",
             "    constexpr int de = CH / P::N, di = CH % P::N;",
             "    __syncthreads();                                  // the previous user of sh is done",
             "    for (size_t w = 2 * (size_t)t; w < total; w += CH) {",
             "        int e1 = e, i1 = i + 1;",
             "        if (i1 == P::N) { i1 = 0; e1++; }",
             "        if (w + 1 < total) {",
             "            const spint2 x = __builtin_nontemporal_load(reinterpret_cast<const spint2*>(aos + w));",
             "            sh[e * SP + i] = x.x; sh[e1 * SP + i1] = x.y;",
             "        } else {",
             "            sh[e * SP + i] = aos[w];",
             "        }",
             "        e += de; i += di;",
             "        if (i >= P::N) { i -= P::N; e++; }",
             "    }",
             "    __syncthreads();",
             "    static_for<0, 2>([&](auto E) {",
             "        const bool live = 2 * (size_t)t + E < cnt;         // lanes beyond the chunk's end compute on zeros (inside the limb contract)",
             "        static_for<0, P::N>([&](auto I) { v[E][I] = live ? sh[(2 * t + E) * SP + I] : 0; });",
             "    });",
             "}",
             "static __device__ __forceinline__ void aos_load(const spint* aos, size_t cnt, spint* sh) {",
             "    const int t = threadIdx.x;",
             "    const size_t total = cnt * (size_t)P::N;",
             "    int e = (2 * t) / P::N, i = (2 * t) % P::N;",
             "    constexpr int de = CH / P::N, di = CH % P::N;",
             "    for (size_t w = 2 * (size_t)t; w < total; w += CH) {",
             "        int e1 = e, i1 = i + 1;",
             "        if (i1 == P::N) { i1 = 0; e1++; }",
             "        if (w + 1 < total) {",
             "            const spint2 x = __builtin_nontemporal_load(reinterpret_cast<const spint2*>(aos + w));",
             "            sh[e * SP + i] = x.x; sh[e1 * SP + i1] = x.y;",
             "        } else {",
             "            sh[e * SP + i] = aos[w];",
             "        }",
             "        e += de; i += di;",
             "        if (i >= P::N) { i -= P::N; e++; }",
             "    }",
             "}",
             "static __device__ __forceinline__ void aos_regs(size_t cnt, const spint* sh, spint (*v)[P::N]) {",
             "    const int t = threadIdx.x;",
             "    static_for<0, 2>([&](auto E) {",
             "        const bool live = 2 * (size_t)t + E < cnt;",
             "        static_for<0, P::N>([&](auto I) { v[E][I] = live ? sh[(2 * t + E) * SP + I] : 0; });",
             "    });",
             "}",
             "static __device__ __forceinline__ void aos_out(spint* aos, size_t cnt, spint* sh, spint (*v)[P::N]) {",
             "    const int t = threadIdx.x;",
             "    const size_t total = cnt * (size_t)P::N;",
             "    __syncthreads();",
             "    static_for<0, 2>([&](auto E) { static_for<0, P::N>([&](auto I) { sh[(2 * t + E) * SP + I] = v[E][I]; }); });",
             "    __syncthreads();",
             "    int e = (2 * t) / P::N, i = (2 * t) % P::N;",
             "    constexpr int de = CH / P::N, di = CH % P::N;",
             "    for (size_t w = 2 * (size_t)t; w < total; w += CH) {",
             "        int e1 = e, i1 = i + 1;",
             "        if (i1 == P::N) { i1 = 0; e1++; }",
             "        if (w + 1 < total) {",
             "            spint2 x; x.x = sh[e * SP + i]; x.y = sh[e1 * SP + i1];",
             "            __builtin_nontemporal_store(x, reinterpret_cast<spint2*>(aos + w));",
             "        } else {",
             "            aos[w] = sh[e * SP + i];",
             "        }",
             "        e += de; i += di;",
             "        if (i >= P::N) { i -= P::N; e++; }",
             "    }",
             "}",
             "__global__ __launch_bounds__(BLOCK) void k_chain_aos(Args A, size_t n) {",
             "    __shared__ spint sh[CH * SP];",
             "    constexpr int EPT = 2;",
             "    const int t = threadIdx.x;",
             "    (void)t;",
             "    for (size_t c0 = (size_t)blockIdx.x * CH; c0 < n; c0 += (size_t)gridDim.x * CH) {",
             "        const size_t cnt = (n - c0 < (size_t)CH) ? n - c0 : (size_t)CH;",
             "        " + " ".join("spint v%d[EPT][P::N];" % i for i in range(nv))]
        if getattr(self, "aos_multi", self.nin * 512 * (N | 1) * 8 <= 65536):
            # one LDS image per input array while they fit 64 KB: every array's loads are issued before the first barrier
            # (measured against one shared image in round 3: X25519 0.577 -> 0.517 ms, X448 1.089 -> 1.038 ms)
            L = [l.replace("__shared__ spint sh[CH * SP];", "__shared__ spint sh[%d * CH * SP];" % self.nin) for l in L]
            L += ["        __syncthreads();"]
            L += ["        aos_load(A.in[%d] + c0 * (size_t)P::N, cnt, sh + %d * CH * SP);" % (i, i) for i in range(self.nin)]
            L += ["        __syncthreads();"]
            L += ["        aos_regs(cnt, sh + %d * CH * SP, v%d);" % (i, i) for i in range(self.nin)]
        else:
            L += ["        aos_in(A.in[%d] + c0 * (size_t)P::N, cnt, sh, v%d);" % (i, i) for i in range(self.nin)]
        L += ["        int s%d[EPT]; static_for<0, EPT>([&](auto E) { s%d[E] = (2 * (size_t)t + E < cnt) ? A.sel[%d][c0 + 2 * (size_t)t + E] : 0; });" % (k, k, k) for k in range(self.nsel)]
        L += ["        " + l for l in votel]
        L += ["        aos_out(A.out[%d] + c0 * (size_t)P::N, cnt, sh, v%d);" % (k, o) for k, o in enumerate(self.outs)]
        L += ["    }", "}"]
        return L

    def _aos_entry(self) -> List[str]:
        if self.params.nlimbs > 14:
            return ['extern "C" int %s(const void* const*, void* const*, size_t, void*) {' % self.aos_symbol,
                    '    set_error("%s: element-major I/O is built for fields of up to 14 limbs; convert with modarith_amd_aos_to_soa");' % self.aos_symbol,
                    "    return (int)hipErrorInvalidValue;", "}", ""]
        return ['extern "C" int %s(const void* const* in, void* const* out, size_t n, void* stream) {' % self.aos_symbol,
                "    // element-major arrays spint x[n][Nlimbs] (16-byte aligned); in[NIN ..): int32 selector arrays as in the _batch form",
                "    if (n == 0) return 0;",
                "    Args A;",
                "    bool al = true;",
                "    for (int i = 0; i < NIN; i++) { A.in[i] = (const spint*)in[i]; al = al && aligned16(in[i]); }",
                "    for (int i = 0; i < NOUT; i++) { A.out[i] = (spint*)out[i]; al = al && aligned16(out[i]); }",
                "    for (int i = 0; i < NSEL; i++) A.sel[i] = (const int*)in[NIN + i];",
                '    if (!al) { set_error("%s: element-major arrays must be 16-byte aligned"); return (int)hipErrorInvalidValue; }' % self.aos_symbol,
                "    const size_t chunks = (n + CH - 1) / CH;",
                "    const size_t cap = (size_t)max_blocks_tiled();",
                "    k_chain_aos<<<(unsigned)(chunks < cap ? chunks : cap), BLOCK, 0, (hipStream_t)stream>>>(A, n);",
                '    return check_launch("%s");' % self.aos_symbol,
                "}", ""]

    # ------------------------------------------------------------------ building the plug-in
    def lib_path(self, plugin_dir: Optional[str] = None) -> str:
        return os.path.join(plugin_dir or _gen.PLUGIN_DIR, "libmodarith_amd_chain_%s_%s.so" % (self.name, self.prime))

    def build(self, plugin_dir: Optional[str] = None, force: bool = False, verbose: bool = False, ept: Optional[int] = None, policy: str = "vote", waves: int = 0) -> "FusedChain":
        from .build import ARCH, FLAGS, HIPCC, _stamp
        d = plugin_dir or _gen.PLUGIN_DIR
        os.makedirs(d, exist_ok=True)
        src_text = self.source(ept, policy, waves)
        base = "chain_%s_%s" % (self.name, self.prime)
        src, obj, meta = (os.path.join(d, base + e) for e in (".hip", ".o", ".json"))
        lib = self.lib_path(d)
        key = hashlib.sha256((" ".join(FLAGS) + "\n" + src_text + "\n" + emit.header_text(self.params) + "\n" + _stamp()).encode()).hexdigest()
        fresh = False
        if not force and os.path.exists(lib) and os.path.exists(meta):
            try:
                fresh = json.load(open(meta)).get("hash") == key
            except (ValueError, OSError):
                fresh = False
        if not fresh:
            if not os.path.exists(HIPCC):
                raise RuntimeError("%s not found: fusing a chain needs the ROCm compiler (there is no CPU path)" % HIPCC)
            emit._write(src, src_text)
            inc = ["-I", os.path.join(HERE, "csrc", "generated"), "-I", os.path.join(HERE, "csrc"), "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", _gen.PLUGIN_DIR, "-I", d]
            if verbose:
                print("[modarith_amd] hipcc %s" % os.path.basename(src), flush=True)
            tmp = ".%d.tmp" % os.getpid()              # process-private names, moved into place when complete (see generate.py)
            subprocess.run([HIPCC] + list(FLAGS) + inc + ["-c", src, "-o", obj + tmp], check=True, timeout=int(os.environ.get("MA_BUILD_TIMEOUT", "1500")))
            subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib + tmp, obj + tmp, "-L", HERE, "-l:libmodarith_amd.so",
                                   "-Wl,-rpath,$ORIGIN/" + os.path.relpath(HERE, d), "-Wl,-rpath," + HERE])
            with open(meta + tmp, "w") as f:
                json.dump({"chain": self.name, "prime": self.prime, "inputs": self.nin, "selectors": self.nsel, "outputs": len(self.outs),
                           "ops": [o[0] for o in self.ops], "symbol": self.symbol, "hash": key}, f, indent=1)
            os.replace(obj + tmp, obj)
            os.replace(lib + tmp, lib)
            os.replace(meta + tmp, meta)
        return FusedChain(self, lib, built=not fresh)


class FusedChain:
    """a built chain: call it with one device batch per input (flat [N, n] or tiled [n / tile, N, tile], all the same shape)"""
    def __init__(self, chain: Chain, lib: str, built: bool):
        self.chain, self.path, self.built = chain, lib, built
        _lib.load()
        self.lib = ctypes.CDLL(lib)
        self.fn = getattr(self.lib, chain.symbol)
        self.fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        self.fn.restype = ctypes.c_int
        self._fields = {}

    def aos(self, *inputs, out: Optional[Sequence] = None):
        """the same chain over element-major device arrays int64 [n, Nlimbs] (`spint x[n][Nlimbs]`, how CPU callers of field.c hold
        elements): no aos_to_soa / soa_to_aos passes around it; selectors follow the element arrays as in __call__"""
        import torch
        ch = self.chain
        N = ch.params.nlimbs
        if len(inputs) != ch.nin + ch.nsel:
            raise ValueError("chain %s takes %d element arrays followed by %d int32 selector arrays" % (ch.name, ch.nin, ch.nsel))
        elems, sels = inputs[:ch.nin], inputs[ch.nin:]
        n = elems[0].shape[0]
        outs = list(out) if out is not None else [torch.empty_like(elems[0]) for _ in ch.outs]
        for t in list(elems) + outs:
            if t.dtype != torch.int64 or t.dim() != 2 or tuple(t.shape) != (n, N) or not t.is_cuda or not t.is_contiguous():
                raise ValueError("expected contiguous int64 device tensors of shape [n, %d]" % N)
        for d in sels:
            if d.dtype != torch.int32 or d.numel() != n or not d.is_cuda or not d.is_contiguous():
                raise ValueError("selectors are contiguous int32 device tensors with one 0/1 entry per element")
        fn = getattr(self.lib, ch.aos_symbol)
        fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
        fn.restype = ctypes.c_int
        ins = (ctypes.c_void_p * (ch.nin + ch.nsel))(*[t.data_ptr() for t in list(elems) + list(sels)])
        ous = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        with torch.cuda.device(elems[0].device):
            _lib.check(fn(ins, ous, n, torch.cuda.current_stream(elems[0].device).cuda_stream), ch.aos_symbol)
        return tuple(outs)

    def __call__(self, *inputs, out: Optional[Sequence] = None, device=None):
        import torch
        from .field import Field
        ch = self.chain
        if len(inputs) != ch.nin + ch.nsel:
            raise ValueError("chain %s takes %d element batches followed by %d int32 selector arrays" % (ch.name, ch.nin, ch.nsel))
        inputs, sels = inputs[:ch.nin], inputs[ch.nin:]
        dev = device if device is not None else inputs[0].device
        F = self._fields.get(dev)
        if F is None:                                  # binding a Field derives the prime's constants: once per device, not per call
            F = self._fields[dev] = Field(ch.prime, dev)
        outs = list(out) if out is not None else [F._out(inputs[0], None) for _ in ch.outs]
        if len(outs) != len(ch.outs):
            raise ValueError("chain %s has %d outputs" % (ch.name, len(ch.outs)))
        n = F._chk(*inputs, *outs)
        for d in sels:
            if d.dtype != torch.int32 or d.numel() != n or not d.is_cuda or not d.is_contiguous() or d.device != F.device:
                raise ValueError("selectors are contiguous int32 device tensors with one 0/1 entry per element")
        ins = (ctypes.c_void_p * (ch.nin + ch.nsel))(*[t.data_ptr() for t in list(inputs) + list(sels)])
        ous = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        with torch.cuda.device(F.device):
            _lib.check(self.fn(ins, ous, n, F._ld(inputs[0]), torch.cuda.current_stream(F.device).cuda_stream), ch.symbol)
        return tuple(outs)


def bench_chain(prime: str = "X25519") -> Chain:
    """the chain bench.py times beside the headline and __graft_entry__.build() pre-builds: z = ((a + b)(a - b))^2, four calls"""
    ch = Chain(prime, "bench_prod")
    u, v = ch.inputs(2)
    ch.output(ch.modsqr(ch.modmul(ch.modadd(u, v), ch.modsub(u, v))))
    return ch


# ---------------------------------------------------------------------------------------------------------------------
# command line: a chain written as text, for consumers that do not otherwise touch Python
#   python -m modarith_amd.fuse X25519 accept "in x, y; t = modadd(x, y); w = modsub(x, y); s = modsqr(modmul(t, w)); out modinv(s)"
def parse(prime: str, name: str, text: str) -> Chain:
    """statements separated by ';':  `in a, b`  (element inputs, in order) | `sel d` (int32 selector inputs) | `v = f(args)` |
    `g2, f2 = modcsw(d, g, f)` | `out expr, ...`.  Expressions nest: modsqr(modmul(a, b)).  Integers are accepted where the
    function takes one (modmli, modnsqr).  Function names are the chain's methods, i.e. field.c's."""
    import ast
    ch = Chain(prime, name)
    env = {}

    def ev(node):
        if isinstance(node, ast.Name):
            if node.id not in env:
                raise ValueError("unknown value %r" % node.id)
            return env[node.id]
        if isinstance(node, ast.Constant) and isinstance(node.value, int):
            return node.value
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub) and isinstance(node.operand, ast.Constant):
            return -node.operand.value
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and not node.keywords:
            fn = node.func.id
            if fn not in _OPS or not hasattr(ch, fn):
                raise ValueError("unknown operation %r (available: %s)" % (fn, ", ".join(sorted(_OPS))))
            return getattr(ch, fn)(*[ev(a) for a in node.args])
        raise ValueError("cannot parse %r" % ast.unparse(node))

    for st in [t.strip() for t in text.split(";") if t.strip()]:
        head, _, rest = st.partition(" ")
        if head == "in":
            for nm in [t.strip() for t in rest.split(",")]:
                env[nm] = ch.input()
        elif head == "sel":
            for nm in [t.strip() for t in rest.split(",")]:
                env[nm] = ch.selector()
        elif head == "out":
            for node in ast.parse("(%s,)" % rest, mode="eval").body.elts:
                ch.output(ev(node))
        else:
            node = ast.parse(st).body[0]
            if not isinstance(node, ast.Assign) or len(node.targets) != 1:
                raise ValueError("expected `name = expression`: %r" % st)
            val = ev(node.value)
            tgt = node.targets[0]
            if isinstance(tgt, ast.Tuple):
                if not isinstance(val, tuple) or len(val) != len(tgt.elts):
                    raise ValueError("%r does not produce %d values" % (st, len(tgt.elts)))
                for t, v in zip(tgt.elts, val):
                    env[t.id] = v
            else:
                env[tgt.id] = val
    return ch


def main(argv: List[str]) -> int:
    args = [a for a in argv if not a.startswith("--")]
    if len(args) != 3:
        print('usage: python -m modarith_amd.fuse <prime or tag> <name> "in a, b; t = modadd(a, b); out modsqr(t)" [--force] [--source]')
        return 2
    try:
        ch = parse(*args)
        if "--source" in argv:
            print(ch.source())
            return 0
        f = ch.build(force="--force" in argv, verbose=True)
    except ValueError as e:
        print(e)
        return 2
    print("%s %s" % ("built" if f.built else "up to date:", f.path))
    print("int %s(const void *const *in /* %d element batches%s */, void *const *out /* %d */, size_t n, size_t ld, void *stream);"
          % (ch.symbol, ch.nin, (", then %d int32 selector arrays" % ch.nsel) if ch.nsel else "", len(ch.outs)))
    print("int %s(const void *const *in, void *const *out, size_t n, void *stream);   /* the same over element-major arrays x[n][Nlimbs] */" % ch.aos_symbol)
    print("HBM bytes per element: %d fused, %d call by call" % (ch.traffic_bytes(), ch.unfused_traffic_bytes()))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main(sys.argv[1:]))
