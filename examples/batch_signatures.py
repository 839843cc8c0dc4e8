"""examples/batch_signatures.py -- the reference's two signature programs, nist256.c (ECDSA over P-256) and ed448.c (EdDSA over
Ed448), rewritten over the BATCHED API: n key pairs / signatures / verifications per call, every curve and group-order operation
on the GPU through the C-ABI (modarith_amd.Field / modarith_amd.Curve), the hashes on the host (hashlib: they are per-message
byte work outside the arithmetic path; the reference's hash.c does the same job one message at a time).

Each function keeps the reference function's name, argument meaning and step order, and cites the lines it follows; what was one
`gel` / `point` / `char[BYTES]` there is a batch here.  Byte strings cross the API as uint8 tensors [n, BYTES].

    python examples/batch_signatures.py          # runs the two main() programs of the reference on a batch (needs a GPU)

tests/test_gpu_signatures.py checks these flows against Python-integer models of ECDSA / EdDSA and the published test vectors.
"""
from __future__ import annotations

import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from modarith_amd.edwards import Curve  # noqa: E402
from modarith_amd.field import Field  # noqa: E402


def _dev(rows) -> torch.Tensor:
    return torch.tensor([list(r) for r in rows], dtype=torch.uint8, device="cuda").contiguous()


def _rows(t: torch.Tensor):
    return [bytes(r) for r in t.cpu().tolist()]


# ------------------------------------------------------------------------------------------------ ECDSA, nist256.c
class Nist256:
    BYTES = 32

    def __init__(self):
        self.C = Curve("NIST256")
        self.G = Field("NIST256Q", tile=None)          # the group-order field (curve.py:324-329)

    def reduce(self, h):
        """nist256.c:122-146: 40 little-endian bytes -> integer mod q as 2^248 x + y (x: top 9 bytes, y: bottom 31)"""
        n = len(h)
        c = self.G.mod2r(8 * (self.BYTES - 1), n)
        y, _ = self.G.modimp(_dev([(bytes(r[:31]) + b"\0")[::-1] for r in h]))
        x, _ = self.G.modimp(_dev([(bytes(r[31:40]) + b"\0" * 23)[::-1] for r in h]))
        self.G.modmul(x, c, x)
        return self.G.modadd(x, y)

    def key_pair(self, compress: bool, prv):
        """NIST256_KEY_PAIR (nist256.c:150-161): ecnXXXgen, ecnXXXmul, ecnXXXget in one kernel"""
        x, y, sign = self.C.mulgen_get(_dev(prv), want_y=not compress)
        xs = _rows(x)
        if compress:
            return [bytes([0x02 + int(s)]) + xb for s, xb in zip(sign.cpu().tolist(), xs)]
        return [b"\x04" + xb + yb for xb, yb in zip(xs, _rows(y))]

    def sign(self, prv, ran, thm):
        """NIST256_SIGN (nist256.c:196-222); ran: 40 random bytes per message, thm: the truncated hash"""
        G = self.G
        e, _ = G.modimp(_dev(thm))
        s, _ = G.modimp(_dev(prv))
        k = self.reduce(ran)
        h = G.modexp(k)
        x, _, _ = self.C.mulgen_get(h, want_y=False)            # ecnXXXgen(&R); ecnXXXmul(h,&R); ecnXXXget(&R,h,NULL)
        kinv = G.modinv(k)
        r, _ = G.modimp(x)
        G.modmul(s, r, s)
        G.modadd(s, e, s)
        G.modmul(s, kinv, s)
        return [a + b for a, b in zip(_rows(G.modexp(r)), _rows(G.modexp(s)))]

    def verify(self, pub, thm, sig):
        """NIST256_VERIFY (nist256.c:226-260) -> list of 0 / 1.  pub: 65-byte (0x04) or 33-byte (0x02 / 0x03) keys, all of one kind"""
        G, B = self.G, self.BYTES
        n = len(sig)
        e, _ = G.modimp(_dev(thm))
        r, r_ok = G.modimp(_dev([s[:B] for s in sig]))
        s, s_ok = G.modimp(_dev([s[B:] for s in sig]))
        ok = (r_ok != 0) & (s_ok != 0) & (G.modis0(r) == 0) & (G.modis0(s) == 0)
        # (rejected records run the remaining steps on whatever they hold, as lanes of a batch do; their verdict is already 0)
        sinv = G.modinv(s)
        v = G.modexp(G.modmul(r, sinv))
        u = G.modexp(G.modmul(sinv, e))
        if pub[0][0] == 0x04:
            Q = self.C.set(None, _dev([p[1:1 + B] for p in pub]), _dev([p[1 + B:] for p in pub]))
        else:
            Q = self.C.set(torch.tensor([p[0] & 1 for p in pub], dtype=torch.int32, device="cuda"), _dev([p[1:] for p in pub]), None)
        x, y, _ = self.C.mulgen2_get(u, v, Q)                    # ecnXXXmul2(u,&G,v,&Q,&Q); ecnXXXget(&Q,rb,NULL)
        inf = (x == 0).all(dim=1) & (y[:, :-1] == 0).all(dim=1) & (y[:, -1] == 1)          # ecnXXXisinf: leaves as (0, 1)
        e2, _ = G.modimp(x)
        ok = ok & ~inf & (G.modcmp(r, e2) != 0)
        assert ok.numel() == n
        return [int(v) for v in ok.cpu().tolist()]


# ------------------------------------------------------------------------------------------------ EdDSA, ed448.c
class Ed448:
    BYTES = 56
    dom4 = b"SigEd448" + b"\0\0"

    def __init__(self):
        self.C = Curve("ED448")
        self.G = Field("ED448Q", tile=None)

    @staticmethod
    def H(data: bytes, olen: int) -> bytes:
        return hashlib.shake_256(data).digest(olen)

    def reduce(self, h):
        """ed448.c:121-153: 114 little-endian bytes -> integer mod q as 2^440 (2^440 x + y) + z"""
        n = len(h)
        G = self.G
        c = G.mod2r(440, n)
        z, _ = G.modimp(_dev([(bytes(r[:55]) + b"\0")[::-1] for r in h]))
        y, _ = G.modimp(_dev([(bytes(r[55:110]) + b"\0")[::-1] for r in h]))
        x, _ = G.modimp(_dev([(bytes(r[110:114]) + b"\0" * 52)[::-1] for r in h]))
        G.modmul(x, c, x)
        G.modadd(x, y, x)
        G.modmul(x, c, x)
        return G.modadd(x, z)

    @staticmethod
    def _clamp(s: bytes) -> bytes:
        b = bytearray(s[:56])
        b[0] &= 0xFC
        b[55] |= 0x80
        return bytes(b)

    def key_pair(self, prv):
        """ED448_KEY_PAIR (ed448.c:167-186); prv: 57 random bytes each -> 57-byte public keys"""
        s = [self._clamp(self.H(p, 56))[::-1] for p in prv]                  # little endian -> big endian
        _, y, sign = self.C.mulgen_get(_dev(s), want_x=False)
        return [yb[::-1] + bytes([int(sg) << 7]) for yb, sg in zip(_rows(y), sign.cpu().tolist())]

    def sign(self, prv, pub, m):
        """ED448_SIGN (ed448.c:191-257)"""
        G, B = self.G, self.BYTES
        if pub is None:
            pub = self.key_pair(prv)
        h = [bytearray(self.H(p, 2 * B + 2)) for p in prv]
        sb = [self._clamp(bytes(x))[::-1] for x in h]
        s, _ = G.modimp(_dev(sb))
        r = self.reduce([self.H(self.dom4 + bytes(x[B + 1:]) + mm, 2 * B + 2) for x, mm in zip(h, m)])
        _, y, sign = self.C.mulgen_get(G.modexp(r), want_x=False)            # ecnXXXmul(h,&R); ecnXXXget(&R,NULL,sig)
        R = [yb[::-1] + bytes([int(sg) << 7]) for yb, sg in zip(_rows(y), sign.cpu().tolist())]
        d = self.reduce([self.H(self.dom4 + Rb + pk + mm, 2 * B + 2) for Rb, pk, mm in zip(R, pub, m)])
        G.modmul(d, s, d)
        G.modadd(d, r, d)
        return [Rb + db[::-1] + b"\0" for Rb, db in zip(R, _rows(G.modexp(d)))]

    def verify(self, pub, m, sig):
        """ED448_VERIFY (ed448.c:261-311) -> list of 0 / 1"""
        G, C, B = self.G, self.C, self.BYTES
        n = len(sig)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")     # noqa: E731
        R = C.set(i32([s[B] >> 7 for s in sig]), None, _dev([s[:B][::-1] for s in sig]))
        ok = C.isinf(R) == 0
        Q = C.set(i32([(p[B] >> 7) & 1 for p in pub]), None, _dev([p[:B][::-1] for p in pub]))
        ok = ok & (C.isinf(Q) == 0)
        buff = _dev([s[B + 1:2 * B + 1][::-1] for s in sig])
        u = self.reduce([self.H(self.dom4 + s[:B + 1] + p + mm, 2 * B + 2) for s, p, mm in zip(sig, pub, m)])
        G.modneg(u, u)
        h = G.modexp(u)
        _, in_range = G.modimp(buff)
        ok = ok & (in_range != 0)
        Gp = C.cof(C.gen(n))
        C.cof(R)
        C.cof(Q)
        Q = C.mul2(buff, Gp, h, Q)
        ok = ok & (C.cmp(R, Q) != 0)
        return [int(v) for v in ok.cpu().tolist()]


def main():
    # nist256.c:264-296
    sk = bytes.fromhex("519b423d715f8b581f4fa8ee59f4771a5b44c8130b4e3eacca54a56dda72b464")
    ran = bytes.fromhex("94a1bbb14b906a61a280f245f9e93c7f3b4a6247824f5d33b9670787642a68deb9670787642a68de")
    msg = bytes.fromhex("44acf6b7e36c1342c2c5897204fe09504e1e2efb1a900377dbc4e7a6a133ec56")
    N = Nist256()
    n = 8
    print("Run test vector (%d lanes)" % n)
    pub = N.key_pair(True, [sk] * n)
    print("public key=", pub[0].hex())
    sig = N.sign([sk] * n, [ran] * n, [msg] * n)
    print("signature= ", sig[0].hex())
    print("Signature is valid" if all(N.verify(pub, [msg] * n, sig)) else "Signature is NOT valid")
    # ed448.c:315-341
    sk = bytes.fromhex("c4eab05d357007c632f3dbb48489924d552b08fe0c353a0d4a1f00acda2c463afbea67c5e8d2877c5e3bc397a659949ef8021e954e0a12274e")
    E = Ed448()
    print("Run RFC8032 test vector (%d lanes)" % n)
    pub = E.key_pair([sk] * n)
    print("public key= ", pub[0].hex())
    sig = E.sign([sk] * n, pub, [b"\x03"] * n)
    print("signature=  ", sig[0].hex())
    print("Signature is valid" if all(E.verify(pub, [b"\x03"] * n, sig)) else "Signature is NOT valid")


if __name__ == "__main__":
    main()
