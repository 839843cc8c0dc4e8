"""Host-side mirror of the reference's curve API (curve.h:13-29) for Edwards curves, batched.

`Ed = Edwards("ED25519")`; a batch of n projective points is a torch int64 tensor [3, Nlimbs, n]
(x, y, z limb-interleaved SoA).  Scalars and coordinates are uint8 [n, Nbytes] big-endian records, as
the reference's `char *` arguments.  Methods keep the reference's names and argument order
(`add(Q, P)` is P += Q) and launch the HIP kernels through the C-ABI on torch's current stream.
"""
from __future__ import annotations

import os

from typing import Optional

import torch

from . import _lib


def _stream(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Edwards:
    """Batched curve API (one object per stream: the window-table workspaces it allocates for `mul`, `mul_get`, ... are reused by the
    next call, so two host threads / streams driving the SAME object concurrently would share them; the C-ABI takes the workspace
    explicitly).  Despite the name it serves every built curve of the reference's curve layer:
    Edwards ("ED25519", "ED448", "NUMS256E", "ED248", "ED376", "ED500") and short Weierstrass ("NIST256", "NIST384", "NIST521", "SECP256K1", "NUMS256W").  `Curve` is an alias."""

    def __init__(self, curve: str, device: Optional[torch.device] = None):
        self.name = curve.lower()
        if self.name in _lib.CURVES:
            self.lib = _lib.load()
            self.N, self.nbytes = _lib.CURVES[self.name]
            self._field = None
        else:
            # a curve made by modarith_amd.generate.generate_curve (the counterpart of adding a curve to curve.py's table)
            from . import generate as _gen
            if not os.path.exists(_gen.curve_plugin_path(self.name)):
                raise ValueError("curve %r is neither built in (%s) nor generated (%s); see modarith_amd.generate.generate_curve"
                                 % (curve, ", ".join(_lib.CURVES), ", ".join(m["curve"] for m in _gen.installed_curves()) or "none"))
            self.lib, self.N, self.nbytes = _lib.load_curve_plugin(self.name)
            self._field = next(m["field"] for m in _gen.installed_curves() if m["curve"].lower() == self.name)
        from .field import normalise_device
        self.device = normalise_device(device)
        self._ws = None
        self._fws = None          # window tables of the fused mul_get kernels (ED448, NIST256)

    # ------------------------------------------------------------------ plumbing
    def empty(self, n: int) -> torch.Tensor:
        return torch.empty((3, self.N, n), dtype=torch.int64, device=self.device)

    def _chk(self, *ps: torch.Tensor) -> int:
        n = ps[0].shape[2]
        for p in ps:
            if p.dtype != torch.int64 or p.dim() != 3 or p.shape[0] != 3 or p.shape[1] != self.N or p.shape[2] != n \
                    or not p.is_cuda or not p.is_contiguous():
                raise ValueError("expected contiguous int64 device tensors of shape [3, %d, n]" % self.N)
            if p.device != self.device:
                raise ValueError("batch on %s, curve bound to %s" % (p.device, self.device))
        return n

    def _bytes(self, b: Optional[torch.Tensor], n: int):
        if b is None:
            return None
        if b.dtype != torch.uint8 or b.shape != (n, self.nbytes) or not b.is_contiguous() or not b.is_cuda or b.device != self.device:
            raise ValueError("expected a contiguous uint8 device tensor [n, %d]" % self.nbytes)
        return b.data_ptr()

    def _call(self, fn: str, *args):
        f = getattr(self.lib, "ecn_%s_%s_batch" % (self.name, fn))
        with torch.cuda.device(self.device):          # the C-ABI launches on the calling thread's current device
            _lib.check(f(*args), "ecn_%s_%s_batch" % (self.name, fn))

    def _scalars(self, e: Optional[torch.Tensor], n: int):
        if e is None:
            raise ValueError("scalar records are required (uint8 [n, %d], big-endian)" % self.nbytes)
        return self._bytes(e, n)

    # ------------------------------------------------------------------ curve.h API, batched
    def inf(self, n: int):
        P = self.empty(n)
        self._call("inf", P.data_ptr(), n, n, _stream(self.device))
        return P

    def gen(self, n: int):
        P = self.empty(n)
        self._call("gen", P.data_ptr(), n, n, _stream(self.device))
        return P

    def cpy(self, Q):
        P = torch.empty_like(Q)
        n = self._chk(Q, P)
        self._call("cpy", Q.data_ptr(), P.data_ptr(), n, n, _stream(self.device))
        return P

    def add(self, Q, P):
        """P += Q"""
        n = self._chk(Q, P)
        self._call("add", Q.data_ptr(), P.data_ptr(), n, n, _stream(self.device))
        return P

    def sub(self, Q, P):
        """P -= Q"""
        n = self._chk(Q, P)
        self._call("sub", Q.data_ptr(), P.data_ptr(), n, n, _stream(self.device))
        return P

    def _un(self, fn, P):
        n = self._chk(P)
        self._call(fn, P.data_ptr(), n, n, _stream(self.device))
        return P

    def limbs_ok(self, P: torch.Tensor) -> torch.Tensor:
        """int32 [n]: 1 where all three coordinates of the point keep the limb budget (every limb < 2^(Radix+2); for the
        2^256-189 field of NUMS256W / NUMS256E: every limb <= (2^64-1)/mm = 2^52.4, the range in which the folded half-limb
        products of their curve kernels agree with the reference) that the outputs of every function of this library keep and the scalar-multiplication kernels rely on; a 0 marks limbs
        fabricated outside the API (the reference's functions "silently overflow" there too, SURVEY 8b, but differently)."""
        from . import curves
        from .field import Field
        self._chk(P)
        up = self.name.upper()
        fname = self._field or (curves.CURVES[up] if up in curves.CURVES else curves.W_CURVES[up]).field
        F = Field(fname, device=self.device)
        return F.modlimbs(P[0]) & F.modlimbs(P[1]) & F.modlimbs(P[2])

    def dbl(self, P): return self._un("dbl", P)
    def neg(self, P): return self._un("neg", P)
    def cof(self, P): return self._un("cof", P)
    def affine(self, P): return self._un("affine", P)

    def mul(self, e: torch.Tensor, P: torch.Tensor):
        """P = e*P for big-endian scalar records e (constant-time fixed window, edwards.c:435-482)"""
        n = self._chk(P)
        need = int(getattr(self.lib, "ecn_%s_mul_workspace_bytes" % self.name)(n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._call("mul", self._scalars(e, n), P.data_ptr(), n, n, self._ws.data_ptr(), self._ws.numel(), _stream(self.device))
        return P

    FUSED = ("ED25519", "ED448", "NIST256", "SECP256K1")       # curves with a fused mul + get kernel (csrc/ed26.h, csrc/ed28.h, csrc/wn26.h)

    def mul_get(self, e: torch.Tensor, P: torch.Tensor, want_x: bool = True, want_y: bool = True):
        """ecnXXXmul followed by ecnXXXget (the reference's call pattern, ed448.c:182-184) in ONE kernel: the affine
        coordinates of e*P as canonical big-endian byte records, and the sign of the omitted coordinate.  P is not
        modified.  Same bytes as mul() + get() for every point on the curve, about twice as fast.  Precondition (all fused
        *_get methods): the coordinate limbs of P keep the limb budget (every limb < 2^(Radix+2): `limbs_ok(P)`), as every
        point produced by this library does; limbs fabricated above it are truncated by the 32-bit re-packing, whereas
        mul() reproduces the reference's 64-bit behaviour for them."""
        if self.name.upper() not in self.FUSED:
            raise ValueError("no fused mul_get kernel for %s (available: %s)" % (self.name, ", ".join(self.FUSED)))
        n = self._chk(P)
        x = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_x else None
        y = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_y else None
        sign = torch.empty(n, dtype=torch.int32, device=self.device)
        need = int(getattr(self.lib, "ecn_%s_mul_get_workspace_bytes" % self.name)(n))
        if need and (self._fws is None or self._fws.numel() < need):
            self._fws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._call("mul_get", self._scalars(e, n), P.data_ptr(), None if x is None else x.data_ptr(), None if y is None else y.data_ptr(),
                   sign.data_ptr(), n, n, self._fws.data_ptr() if need else None, need, _stream(self.device))
        return x, y, sign

    FUSEDG = ("NIST256", "SECP256K1", "ED25519", "ED448")      # curves with a fused gen + mul + get kernel (fixed-base table)

    def mulgen_get(self, e: torch.Tensor, want_x: bool = True, want_y: bool = True):
        """ecnXXXgen, ecnXXXmul, ecnXXXget (the opening of key generation and signing, nist256.c:150-161, 214-222, ed448.c:167-184, 196-199) in ONE
        kernel: the affine coordinates of e*G as canonical big-endian byte records and the sign of the omitted coordinate.
        Same bytes as get(mul(e, gen(n))); no doublings (precomputed multiples of G), about four times the rate of mul_get."""
        if self.name.upper() not in self.FUSEDG:
            raise ValueError("no fused mulgen_get kernel for %s (available: %s)" % (self.name, ", ".join(self.FUSEDG)))
        if e is None or e.dim() != 2:
            raise ValueError("scalar records are required (uint8 [n, %d], big-endian)" % self.nbytes)
        n = e.shape[0]
        x = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_x else None
        y = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_y else None
        sign = torch.empty(n, dtype=torch.int32, device=self.device)
        self._call("mulgen_get", self._scalars(e, n), None if x is None else x.data_ptr(), None if y is None else y.data_ptr(),
                   sign.data_ptr(), n, _stream(self.device))
        return x, y, sign

    FUSEDG2 = ("NIST256", "SECP256K1", "ED25519", "ED448")      # curves with a fused gen + mul2 + get kernel (e*G + f*Q)

    def mulgen2_get(self, e, f, Q, want_x: bool = True, want_y: bool = True):
        """ecnXXXgen, ecnXXXmul2(e, G, f, Q, R), ecnXXXget (signature verification, nist256.c:251-256, ed448.c:305) in ONE kernel:
        the affine coordinates of e*G + f*Q as canonical big-endian byte records.  Q is not modified.  Same bytes as
        mul2_get(e, gen(n), f, Q); the generator part runs on the fixed-base table."""
        if self.name.upper() not in self.FUSEDG2:
            raise ValueError("no fused mulgen2_get kernel for %s (available: %s)" % (self.name, ", ".join(self.FUSEDG2)))
        n = self._chk(Q)
        x = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_x else None
        y = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_y else None
        sign = torch.empty(n, dtype=torch.int32, device=self.device)
        need = int(getattr(self.lib, "ecn_%s_mulgen2_get_workspace_bytes" % self.name)(n))
        if need and (self._fws is None or self._fws.numel() < need):
            self._fws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._call("mulgen2_get", self._scalars(e, n), self._scalars(f, n), Q.data_ptr(),
                   None if x is None else x.data_ptr(), None if y is None else y.data_ptr(), sign.data_ptr(), n, n,
                   self._fws.data_ptr() if need else None, need, _stream(self.device))
        return x, y, sign

    FUSED2 = ("ED25519", "ED448", "NIST256", "SECP256K1")      # curves with a fused mul2 + get kernel

    def mul2_get(self, e, P, f, Q, want_x: bool = True, want_y: bool = True):
        """ecnXXXmul2 followed by ecnXXXget (the verification pattern, ed448.c:305) in ONE kernel: the affine coordinates of
        e*P + f*Q as canonical big-endian byte records and the sign of the omitted coordinate.  P, Q are not modified."""
        if self.name.upper() not in self.FUSED2:
            raise ValueError("no fused mul2_get kernel for %s (available: %s)" % (self.name, ", ".join(self.FUSED2)))
        n = self._chk(P, Q)
        x = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_x else None
        y = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_y else None
        sign = torch.empty(n, dtype=torch.int32, device=self.device)
        need = int(getattr(self.lib, "ecn_%s_mul2_get_workspace_bytes" % self.name)(n))
        if need and (self._fws is None or self._fws.numel() < need):
            self._fws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._call("mul2_get", self._scalars(e, n), P.data_ptr(), self._scalars(f, n), Q.data_ptr(),
                   None if x is None else x.data_ptr(), None if y is None else y.data_ptr(), sign.data_ptr(), n, n,
                   self._fws.data_ptr() if need else None, need, _stream(self.device))
        return x, y, sign

    def _workspace(self, n: int):
        need = int(getattr(self.lib, "ecn_%s_mul_workspace_bytes" % self.name)(n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def mul2(self, e, P, f, Q, exact: bool = False):
        """R = e*P + f*Q (ecnXXXmul2, edwards.c:486-510); returns a new batch.  Default: two fixed-window multiplications sharing
        their doublings (constant time, no lane divergence) -- the reference's point in another projective representative.
        exact=True: the reference's own walk over its joint sparse form -- the reference's limbs, variable time, 4-18 % slower (docs/curve_layer.md 4.5)."""
        n = self._chk(P, Q)
        R = torch.empty_like(P)
        ws = self._workspace(n)
        self._call("mul2_exact" if exact else "mul2", self._scalars(e, n), P.data_ptr(), self._scalars(f, n), Q.data_ptr(), R.data_ptr(), n, n,
                   ws.data_ptr(), ws.numel(), _stream(self.device))
        return R

    def ran(self, r: int, P):
        n = self._chk(P)
        self._call("ran", int(r), P.data_ptr(), n, n, _stream(self.device))
        return P

    def cmp(self, P, Q):
        n = self._chk(P, Q)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        self._call("cmp", P.data_ptr(), Q.data_ptr(), out.data_ptr(), n, n, _stream(self.device))
        return out

    def isinf(self, P):
        n = self._chk(P)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        self._call("isinf", P.data_ptr(), out.data_ptr(), n, n, _stream(self.device))
        return out

    def set(self, s: Optional[torch.Tensor], x: Optional[torch.Tensor], y: Optional[torch.Tensor]):
        """ecnXXXset: from (x, y), or from x and the sign s of y, or from y and the sign s of x;
        off-curve input gives the point at infinity (edwards.c:246-366).  s: int32 [n] or None."""
        ref = x if x is not None else y
        n = ref.shape[0]
        P = self.empty(n)
        sp = None
        if s is not None:
            if s.dtype != torch.int32 or s.numel() != n or not s.is_cuda:
                raise ValueError("s must be an int32 device tensor [n]")
            sp = s.data_ptr()
        self._call("set", sp, self._bytes(x, n), self._bytes(y, n), P.data_ptr(), n, n, _stream(self.device))
        return P

    def get(self, P, want_x: bool = True, want_y: bool = True):
        """ecnXXXget: makes P affine in place; returns (x bytes | None, y bytes | None, sign of the omitted
        coordinate as int32 [n])."""
        n = self._chk(P)
        x = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_x else None
        y = torch.empty((n, self.nbytes), dtype=torch.uint8, device=self.device) if want_y else None
        sign = torch.empty(n, dtype=torch.int32, device=self.device)
        self._call("get", P.data_ptr(), None if x is None else x.data_ptr(), None if y is None else y.data_ptr(),
                   sign.data_ptr(), n, n, _stream(self.device))
        return x, y, sign


Curve = Edwards
