"""Curve constants for the layers built on the field path: the table of curve.py:85-105
(Edwards curves a*x^2 + y^2 = 1 + d*x^2*y^2) and its conversion to internal-form limbs
(curve.py:244-298: plain limbs for a pseudo-Mersenne field, value*R mod p for a Montgomery field; a
constant below 2^28 in magnitude stays a C int)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

from .params import FieldParams, derive


@dataclass
class EdwardsCurve:
    name: str               # e.g. "ED25519"
    field: str              # built field prime carrying it ("X25519" | "X448")
    a: int                  # +1 or -1
    d: int                  # curve constant B of curve.py (may be negative and small)
    cof: int                # log2 of the cofactor
    order: int              # prime group order q
    gx: int
    gy: int
    fp: Optional[FieldParams] = None

    @property
    def small_b(self) -> bool:
        return abs(self.d) < (1 << 28)      # curve.py:235-240

    @property
    def small_x(self) -> bool:
        return abs(self.gx) < (1 << 28)     # curve.py:239-240: generator decompressed from CONSTANT_X

    def internal(self, v: int) -> List[int]:
        """field element -> internal-form limbs (top limb masked: the value is canonical)"""
        fp = self.fp
        if fp.montgomery:
            v = v * fp.R % fp.p
        return fp.to_limbs(v % fp.p, masked_top=True)


CURVES = {
    "ED25519": EdwardsCurve(
        "ED25519", "X25519", -1,
        0x52036CEE2B6FFE738CC740797779E89800700A4D4141D8AB75EB4DCA135978A3, 3,
        0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED,
        0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A,
        0x6666666666666666666666666666666666666666666666666666666666666658),
    "ED448": EdwardsCurve(
        "ED448", "X448", 1, -39081, 2,
        (2**448 - 2**224 - 1 + 1 - 28312320572429821613362531907042076847709625476988141958474579766324) // 4,
        0x4f1970c66bed0ded221d15a622bf36da9e146570470f1767ea6de324a3d3a46412ae1af72ab66511433b80e18b00938e2626a82bc70cc05e,
        0x693f46716eb6bc248876203756c9c7624bea73736ca3984087789c1e05a0c2d73ad3ff1ce67c39c4fdbd132c4ed7c8ad9808795bf230fa14),
    # curve.py:137-145: x^2 + y^2 = 1 - 15342 x^2 y^2 over 2^256-189, generator from CONSTANT_X = 34 (y of even sign)
    "NUMS256E": EdwardsCurve(
        "NUMS256E", "NUMS256W", 1, -15342, 2,
        0x4000000000000000000000000000000041955AA52F59439B1A47B190EEDD4AF5,
        34, 0),
    # curve.py:107-135: x^2 + y^2 = 1 + d x^2 y^2 with small negative d and small generator x
    "ED248": EdwardsCurve("ED248", "ED248", 1, -107431, 2,
                          0x13FFFFFFFFFFFFFFFFFFFFFFFFFFFFFF098677E8D0D856DA332BA970DCFDEA1, 4, 0),
    "ED376": EdwardsCurve("ED376", "ED376", 1, -66524, 2,
                          0x104000000000000000000000000000000000000000000000303A69B3514879CD109A98F29F0D04F09F855D4F3C6A7037, 2, 0),
    "ED500": EdwardsCurve("ED500", "ED500", 1, -105355, 2,
                          0x6C00000000000000000000000000000000000000000000000000000000000002C8858DA0CB07C5ABCADABC1BEE86F8C9101174D8A115AD57E5F0228C2D0871, 6, 0),
}


@dataclass
class WeierstrassCurve:
    name: str
    field: str
    a: int                  # -3 or 0 (weierstrass.c handles these two)
    b: int
    order: int
    gx: int
    gy: int
    fp: Optional[FieldParams] = None

    @property
    def small_b(self) -> bool:
        return abs(self.b) < (1 << 28)      # curve.py:235-238: b stays the C int CONSTANT_B

    @property
    def small_x(self) -> bool:
        return abs(self.gx) < (1 << 28)     # curve.py:239-240: generator decompressed from CONSTANT_X

    def internal(self, v: int) -> List[int]:
        fp = self.fp
        if fp.montgomery:
            v = v * fp.R % fp.p
        return fp.to_limbs(v % fp.p, masked_top=True)


W_CURVES = {
    # curve.py:157-166
    "NIST256": WeierstrassCurve(
        "NIST256", "NIST256", -3,
        0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b,
        0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551,
        0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296,
        0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5),
    # curve.py:168-177
    "NIST384": WeierstrassCurve(
        "NIST384", "NIST384", -3,
        27580193559959705877849011840389048093056905856361568521428707301988689241309860865136260764883745107765439761230575,
        39402006196394479212279040100143613805079739270465446667948293404245721771496870329047266088258938001861606973112319 + 1
        - 1388124618062372383606759648309780106643088307173319169677,
        0xaa87ca22be8b05378eb1c71ef320ad746e1d3b628ba79b9859f741e082542a385502f25dbf55296c3a545e3872760ab7,
        0x3617de4a96262c6f5d9e98bf9292dc29f8f41dbd289a147ce9da3113b5f0b8c00a60b1ce1d7e819d7a431d7c90ea0e5f),
    # curve.py:179-188
    "NIST521": WeierstrassCurve(
        "NIST521", "NIST521", -3,
        0x51953EB9618E1C9A1F929A21A0B68540EEA2DA725B99B315F3B8B489918EF109E156193951EC7E937B1652C0BD3BB1BF073573DF883D2C34F1EF451FD46B503F00,
        0x1fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffa51868783bf2f966b7fcc0148f709a5d03bb5c9b8899c47aebb6fb71e91386409,
        0xC6858E06B70404E9CD9E3ECB662395B4429C648139053FB521F828AF606B4D3DBAA14B5E77EFE75928FE1DC127A2FFA8DE3348B3C1856A429BF97E7E31C2E5BD66,
        0x11839296A789A3BC0045C8A5FB42C7D1BD998F54449579B446817AFBD17273E662C97EE72995EF42640C550B9013FAD0761353C7086A272C24088BE94769FD16650),
    # curve.py:190-198; at 64 bits over pseudo.py's field
    "SECP256K1": WeierstrassCurve(
        "SECP256K1", "SECP256K1", 0, 7,
        0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
        0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
        0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8),
    # curve.py:147-155: small b and small generator x (y is recovered, even sign)
    "NUMS256W": WeierstrassCurve(
        "NUMS256W", "NUMS256W", -3, 152961,
        0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE43C8275EA265C6020AB20294751A825,
        2, 0),
}


def wcurve(name: str) -> WeierstrassCurve:
    c = W_CURVES[name]
    if c.fp is None:
        c.fp = derive(c.field)
    return c


def curve(name: str) -> EdwardsCurve:
    c = CURVES[name]
    if c.fp is None:
        c.fp = derive(c.field)
    return c
