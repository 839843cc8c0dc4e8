"""The callers of the curve layer: the reference's signature programs nist256.c (ECDSA / P-256) and ed448.c (EdDSA / Ed448) run
as batches over the C-ABI (examples/batch_signatures.py keeps the reference's functions, step for step) and are checked against
Python-integer models of the two schemes written here from FIPS 186-5 / RFC 8032, the FIPS 186 P-256 / SHA-256 signature vector
and the RFC 8032 Ed448 vectors (the "1 octet" one is the vector ed448.c:315-341 runs).  Valid signatures, tampered messages,
r / s = 0 or out of range, off-curve and small-order inputs -- the verdicts must be the reference's (nist256.c:226-260,
ed448.c:261-311)."""
import hashlib
import os
import random
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


# ------------------------------------------------------------------ Python-integer models (test infrastructure)
class P256:
    p = 2**256 - 2**224 + 2**192 + 2**96 - 1
    q = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
    b = 0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b
    G = (0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296, 0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5)

    @classmethod
    def add(cls, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        p = cls.p
        if P[0] == Q[0]:
            if (P[1] + Q[1]) % p == 0:
                return None
            lam = (3 * P[0] * P[0] - 3) * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x = (lam * lam - P[0] - Q[0]) % p
        return x, (lam * (P[0] - x) - P[1]) % p

    @classmethod
    def mul(cls, k, P):
        R = None
        while k:
            if k & 1:
                R = cls.add(R, P)
            P = cls.add(P, P)
            k >>= 1
        return R

    @classmethod
    def sign(cls, d, k, e):
        r = cls.mul(k, cls.G)[0] % cls.q
        return r, pow(k, -1, cls.q) * (e + r * d) % cls.q

    @classmethod
    def verify(cls, Q, e, r, s):
        if not (0 < r < cls.q and 0 < s < cls.q) or Q is None:
            return 0
        w = pow(s, -1, cls.q)
        R = cls.add(cls.mul(e * w % cls.q, cls.G), cls.mul(r * w % cls.q, Q))
        return int(R is not None and R[0] % cls.q == r)


class E448:
    p = 2**448 - 2**224 - 1
    d = -39081 % p
    q = 2**446 - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d
    G = (224580040295924300187604334099896036246789641632564134246125461686950415467406032909029192869357953282578032075146446173674602635247710,
         298819210078481492676017930443930673437544040154080242095928241372331506189835876003536878655418784733982303233503462500531545062832660)

    @classmethod
    def add(cls, P, Q):
        p = cls.p
        x1, y1 = P
        x2, y2 = Q
        t = cls.d * x1 * x2 * y1 * y2 % p
        return (x1 * y2 + x2 * y1) * pow(1 + t, -1, p) % p, (y1 * y2 - x1 * x2) * pow(1 - t, -1, p) % p

    @classmethod
    def mul(cls, k, P):
        R = (0, 1)
        while k:
            if k & 1:
                R = cls.add(R, P)
            P = cls.add(P, P)
            k >>= 1
        return R

    @classmethod
    def enc(cls, P):
        return P[1].to_bytes(56, "little") + bytes([(P[0] & 1) << 7])

    @classmethod
    def dec(cls, b):
        p = cls.p
        y, sign = int.from_bytes(b[:56], "little"), b[56] >> 7
        if y >= p or (b[56] & 0x7f):
            return None
        u, v = (y * y - 1) % p, (cls.d * y * y - 1) % p
        x = pow(u, 3, p) * v % p * pow(pow(u, 5, p) * pow(v, 3, p) % p, (p - 3) // 4, p) % p
        if v * x * x % p != u:
            return None
        if x == 0 and sign:
            return None
        return ((p - x) if (x & 1) != sign else x), y

    @staticmethod
    def H(data, n=114):
        return hashlib.shake_256(data).digest(n)

    @classmethod
    def secret(cls, prv):
        h = bytearray(cls.H(prv))
        h[0] &= 0xFC
        h[55] |= 0x80
        h[56] = 0
        return int.from_bytes(h[:57], "little"), bytes(h[57:])

    @classmethod
    def public(cls, prv):
        return cls.enc(cls.mul(cls.secret(prv)[0], cls.G))

    @classmethod
    def sign(cls, prv, m):
        s, prefix = cls.secret(prv)
        A = cls.public(prv)
        dom = b"SigEd448\0\0"
        r = int.from_bytes(cls.H(dom + prefix + m), "little") % cls.q
        R = cls.enc(cls.mul(r, cls.G))
        k = int.from_bytes(cls.H(dom + R + A + m), "little") % cls.q
        return R + ((r + k * s) % cls.q).to_bytes(57, "little")

    @classmethod
    def verify(cls, A, m, sig):
        """the reference's verdict (ed448.c:261-311): cofactored equation [4][S]B = [4]R + [4][k]A, S < q, the 57th byte of S
        not inspected, R / A decoded as ecnXXXset does (a non-canonical y >= p is imported mod p by modimp's contract -- not
        exercised here), decoding failure or the neutral element as R / A -> 0"""
        Rp, Ap = cls.dec(sig[:57]), cls.dec(A)
        if Rp is None or Ap is None or Rp == (0, 1) or Ap == (0, 1):
            return 0
        S = int.from_bytes(sig[57:113], "little")
        if S >= cls.q:
            return 0
        k = int.from_bytes(cls.H(b"SigEd448\0\0" + sig[:57] + A + m), "little") % cls.q
        lhs = cls.mul(4 * S, cls.G)
        rhs = cls.add(cls.mul(4, Rp), cls.mul(4 * k % cls.q, Ap))
        return int(lhs == rhs)


@pytest.fixture(scope="module")
def flows():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import batch_signatures as bs
    return bs


# ------------------------------------------------------------------ ECDSA P-256 (nist256.c)
def test_ecdsa_p256_batch_against_the_integer_model(flows):
    N = flows.Nist256()
    rng = random.Random(186)
    n = 48
    prv = [rng.randrange(1, P256.q).to_bytes(32, "big") for _ in range(n)]
    prv[0] = (1).to_bytes(32, "big")
    prv[1] = (P256.q - 1).to_bytes(32, "big")
    ran = [rng.getrandbits(320).to_bytes(40, "little") for _ in range(n)]
    ran[2] = (1).to_bytes(40, "little")                                     # k = 1: R = G
    ran[3] = (P256.q - 1).to_bytes(40, "little")
    ran[4] = ((1 << 320) - 1).to_bytes(40, "little")
    msgs = [rng.randbytes(rng.randrange(0, 200)) for _ in range(n)]
    thm = [hashlib.sha256(m).digest() for m in msgs]
    thm[5] = b"\xff" * 32                                                   # e >= q: modimp reduces it once
    thm[6] = b"\0" * 32

    pubc = N.key_pair(True, prv)
    pubu = N.key_pair(False, prv)
    for d, pc, pu in zip(prv, pubc, pubu):
        Q = P256.mul(int.from_bytes(d, "big"), P256.G)
        assert pu == b"\x04" + Q[0].to_bytes(32, "big") + Q[1].to_bytes(32, "big")
        assert pc == bytes([2 + (Q[1] & 1)]) + Q[0].to_bytes(32, "big")

    sig = N.sign(prv, ran, thm)
    for d, k40, e32, sg in zip(prv, ran, thm, sig):
        k = int.from_bytes(k40, "little") % P256.q
        r, s = P256.sign(int.from_bytes(d, "big"), k, int.from_bytes(e32, "big") % P256.q)
        assert sg == r.to_bytes(32, "big") + s.to_bytes(32, "big")

    assert N.verify(pubu, thm, sig) == [1] * n
    assert N.verify(pubc, thm, sig) == [1] * n

    # every way a verification must fail, one lane each, valid lanes in between
    bad_sig, bad_thm, bad_pub, want = list(sig), list(thm), list(pubu), [1] * n
    q = P256.q
    def put(i, sg=None, th=None, pk=None):
        if sg is not None:
            bad_sig[i] = sg
        if th is not None:
            bad_thm[i] = th
        if pk is not None:
            bad_pub[i] = pk
        Qb = bad_pub[i]
        Qp = (int.from_bytes(Qb[1:33], "big"), int.from_bytes(Qb[33:], "big"))
        on = (Qp[1] ** 2 - (Qp[0] ** 3 - 3 * Qp[0] + P256.b)) % P256.p == 0 and Qp[0] < P256.p and Qp[1] < P256.p
        r, s = int.from_bytes(bad_sig[i][:32], "big"), int.from_bytes(bad_sig[i][32:], "big")
        want[i] = P256.verify(Qp if on else None, int.from_bytes(bad_thm[i], "big") % q, r, s) if on else 0
    put(1, th=hashlib.sha256(b"another message").digest())
    put(3, sg=b"\0" * 32 + sig[3][32:])                                     # r = 0
    put(5, sg=sig[5][:32] + b"\0" * 32)                                     # s = 0
    put(7, sg=q.to_bytes(32, "big") + sig[7][32:])                          # r = q: out of range
    put(9, sg=sig[9][:32] + (q + 5).to_bytes(32, "big"))                    # s > q
    put(11, sg=sig[11][:32] + ((int.from_bytes(sig[11][32:], "big") + 1) % q).to_bytes(32, "big"))
    put(13, sg=sig[13][:32] + (q - int.from_bytes(sig[13][32:], "big")).to_bytes(32, "big"))     # (r, -s): valid in ECDSA
    put(15, pk=pubu[16])                                                    # somebody else's key
    put(17, pk=b"\x04" + pubu[17][1:33] + ((int.from_bytes(pubu[17][33:], "big") + 1) % P256.p).to_bytes(32, "big"))   # off the curve
    put(19, sg=b"\xff" * 64)
    got = N.verify(bad_pub, bad_thm, bad_sig)
    assert got == want and want[13] == 1 and sum(want) == n - 9, (got, want)


def test_ecdsa_p256_fips_186_signature_vector(flows):
    """FIPS 186-4 SigGen, P-256 / SHA-256, first vector (the private key and message hash of nist256.c:266-268): with the
    vector's k handed over as the canonical nonce -- 40 little-endian bytes of the same integer, so that reduce() returns it --
    NIST256_SIGN must produce the published (R, S), and the Python model must agree with both"""
    N = flows.Nist256()
    d = bytes.fromhex("519b423d715f8b581f4fa8ee59f4771a5b44c8130b4e3eacca54a56dda72b464")
    k = int("94a1bbb14b906a61a280f245f9e93c7f3b4a6247824f5d33b9670787642a68de", 16)
    e = bytes.fromhex("44acf6b7e36c1342c2c5897204fe09504e1e2efb1a900377dbc4e7a6a133ec56")
    R = "f3ac8061b514795b8843e3d6629527ed2afd6b1f6a555a7acabb5e6f79c8c2ac"
    S = "8bf77819ca05a6b2786c76262bf7371cef97b218e96f175a3ccdda2acc058903"
    Qx = "1ccbe91c075fc7f4f033bfa248db8fccd3565de94bbfb12f3c59ff46c271bf83"
    Qy = "ce4014c68811f9a21a1fdb2c0e6113e06db7ca93b7404e78dc7ccd5ca89a4ca9"
    assert hashlib.sha256(bytes.fromhex(
        "5905238877c77421f73e43ee3da6f2d9e2ccad5fc942dcec0cbd25482935faaf416983fe165b1a045ee2bcd2e6dca3bdf46c4310a7461f9a37960ca672d3feb5473e253605fb1ddfd28065b53cb5858a8ad28175bf9bd386a5e471ea7a65c17cc934a9d791e91491eb3754d03799790fe2d308d16146d5c9b0d0debd97d79ce8"
    )).digest() == e
    r, s = P256.sign(int.from_bytes(d, "big"), k, int.from_bytes(e, "big"))
    assert (f"{r:064x}", f"{s:064x}") == (R, S)
    pub = N.key_pair(False, [d] * 4)
    assert pub[0].hex() == "04" + Qx + Qy
    sig = N.sign([d] * 4, [k.to_bytes(40, "little")] * 4, [e] * 4)
    assert sig[0].hex() == R + S and len(set(sig)) == 1
    assert N.verify(pub, [e] * 4, sig) == [1] * 4
    # nist256.c's own main(): its 40-byte nonce string read the way reduce() reads it (little-endian), against the model
    ran = bytes.fromhex("94a1bbb14b906a61a280f245f9e93c7f3b4a6247824f5d33b9670787642a68deb9670787642a68de")
    r2, s2 = P256.sign(int.from_bytes(d, "big"), int.from_bytes(ran, "little") % P256.q, int.from_bytes(e, "big"))
    sig2 = N.sign([d], [ran], [e])
    assert sig2[0] == r2.to_bytes(32, "big") + s2.to_bytes(32, "big")
    assert N.verify(N.key_pair(True, [d]), [e], sig2) == [1]


# ------------------------------------------------------------------ EdDSA Ed448 (ed448.c)
RFC8032_ED448 = [   # (secret key, public key, message, signature): RFC 8032 section 7.4, "Blank" and "1 octet"
    ("6c82a562cb808d10d632be89c8513ebf6c929f34ddfa8c9f63c9960ef6e348a3528c8a3fcc2f044e39a3fc5b94492f8f032e7549a20098f95b",
     "5fd7449b59b461fd2ce787ec616ad46a1da1342485a70e1f8a0ea75d80e96778edf124769b46c7061bd6783df1e50f6cd1fa1abeafe8256180",
     "",
     "533a37f6bbe457251f023c0d88f976ae2dfb504a843e34d2074fd823d41a591f2b233f034f628281f2fd7a22ddd47d7828c59bd0a21bfd3980"
     "ff0d2028d4b18a9df63e006c5d1c2d345b925d8dc00b4104852db99ac5c7cdda8530a113a0f4dbb61149f05a7363268c71d95808ff2e652600"),
    ("c4eab05d357007c632f3dbb48489924d552b08fe0c353a0d4a1f00acda2c463afbea67c5e8d2877c5e3bc397a659949ef8021e954e0a12274e",
     "43ba28f430cdff456ae531545f7ecd0ac834a55d9358c0372bfa0c6c6798c0866aea01eb00742802b8438ea4cb82169c235160627b4c3a9480",
     "03",
     "26b8f91727bd62897af15e41eb43c377efb9c610d48f2335cb0bd0087810f4352541b143c4b981b7e18f62de8ccdf633fc1bf037ab7cd77980"
     "5e0dbcc0aae1cbcee1afb2e027df36bc04dcecbf154336c19f0af7e0a6472905e799f1953d2a0ff3348ab21aa4adafd1d234441cf807c03a00"),
]


def test_eddsa_ed448_rfc8032_vectors(flows):
    E = flows.Ed448()
    for sk, pk, m, sg in RFC8032_ED448:
        sk, pk, m, sg = bytes.fromhex(sk), bytes.fromhex(pk), bytes.fromhex(m), bytes.fromhex(sg)
        assert E448.public(sk) == pk and E448.sign(sk, m) == sg and E448.verify(pk, m, sg) == 1       # the model reproduces the RFC
        assert E.key_pair([sk] * 3) == [pk] * 3
        assert E.sign([sk] * 3, None, [m] * 3) == [sg] * 3
        assert E.verify([pk] * 3, [m] * 3, [sg] * 3) == [1] * 3


def test_eddsa_ed448_batch_against_the_integer_model(flows):
    E = flows.Ed448()
    rng = random.Random(8032)
    n = 24
    prv = [rng.randbytes(57) for _ in range(n)]
    msgs = [rng.randbytes(rng.randrange(0, 120)) for _ in range(n)]
    pub = E.key_pair(prv)
    assert pub == [E448.public(p) for p in prv]
    sig = E.sign(prv, pub, msgs)
    assert sig == [E448.sign(p, m) for p, m in zip(prv, msgs)]
    assert E.sign(prv, None, msgs) == sig                                    # pub == NULL: derived inside (ed448.c:207-209)
    assert E.verify(pub, msgs, sig) == [1] * n

    bad_sig, bad_msg, bad_pub = list(sig), list(msgs), list(pub)
    q = E448.q
    S = lambda i: int.from_bytes(sig[i][57:113], "little")                  # noqa: E731
    bad_msg[1] = msgs[1] + b"!"
    bad_sig[3] = sig[3][:57] + ((S(3) + 1) % q).to_bytes(56, "little") + b"\0"
    bad_sig[5] = sig[5][:57] + (S(5) + q).to_bytes(56, "little") + b"\0"    # S + q: same residue, out of range
    bad_sig[7] = sig[6][:57] + sig[7][57:]                                   # R of another signature
    bad_pub[9] = pub[10]
    bad_sig[11] = (1).to_bytes(56, "little") + b"\0" + sig[11][57:]          # R = the neutral element (y = 1)
    bad_pub[13] = (1).to_bytes(56, "little") + b"\0"                         # A = the neutral element
    ynon = next(y for y in range(2, 100) if E448.dec(y.to_bytes(56, "little") + b"\0") is None)
    bad_sig[15] = ynon.to_bytes(56, "little") + b"\0" + sig[15][57:]         # R: y with no x on the curve
    bad_pub[17] = ynon.to_bytes(56, "little") + b"\0"
    bad_sig[19] = (E448.p - 1).to_bytes(56, "little") + b"\0" + sig[19][57:]  # R = (0, -1): order 2, killed by the cofactor -> compares as O
    bad_sig[21] = sig[21][:56] + bytes([sig[21][56] ^ 0x80]) + sig[21][57:]  # the other x
    want = [E448.verify(a, m, s) for a, m, s in zip(bad_pub, bad_msg, bad_sig)]
    got = E.verify(bad_pub, bad_msg, bad_sig)
    assert got == want and sum(want) == n - 11, (got, want)


def test_example_program_prints_the_reference_transcript():
    """python examples/batch_signatures.py: the two main() programs of the reference (nist256.c:264-296, ed448.c:315-349) on 8
    lanes -- the RFC 8032 public key and signature and both 'Signature is valid' lines"""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "batch_signatures.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    assert out.count("Signature is valid") == 2 and "NOT valid" not in out
    assert RFC8032_ED448[1][1] in out and RFC8032_ED448[1][3] in out


def test_c_example_ecdsa_verify_batch(tmp_path):
    """examples/ecdsa_verify_batch.c: NIST256_VERIFY (nist256.c:226-260) for n signatures through the plain C-ABI -- the FIPS 186
    vector in every fourth lane, the reference's three rejections (other message, s = 0, r out of range) in the others"""
    import subprocess
    exe = str(tmp_path / "ecdsa_verify_batch")
    subprocess.check_call(["gcc", "-O2", os.path.join(ROOT, "examples", "ecdsa_verify_batch.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "modarith_amd"), "-l:libmodarith_amd.so", "-Wl,-rpath," + os.path.join(ROOT, "modarith_amd"), "-o", exe])
    for n in (4096, 10000, 7):
        p = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        out = p.stdout.splitlines()
        assert out[-2] == "%d signatures, %d valid, %d verdicts as expected" % (n, (n + 3) // 4, n) and out[-1].endswith("as the reference decides")
