"""Per-prime parameter driver: prime -> radix, limb count and every constant the HIP kernels need.

This is the build's counterpart of the parameter blocks of the reference generators
(pseudo.py:124-147 + 1561-1678 for 2^n-c primes, monty.py:151-298 + 2129-2253 for Montgomery-form
primes).  It derives the same quantities from first principles so that the limb layout and the
non-canonical (< 2p) results of the kernels are bit-identical to the reference's generated field.c:

  pseudo-Mersenne : n, Radix, Nlimbs, xcess, m, mm = m*2^xcess, TW, and the variant flags
                    overflow / fred ("tighter reduction") / EPM / carry_on
  Montgomery      : Radix (with the "excess >= 2 or virtual limb" rule), signed prime limbs ppw[]
                    (2^Radix-1 rewritten to -1 with a carry), virtual-limb flag E, R, ndash,
                    R^2 mod p limbs (the nres constant), trinomial index
  both            : Nbytes, PM1D2 (2-adicity of p-1), PE (progenitor exponent), root of unity.

tests/test_params.py checks every value against the macro block / constants captured from the
reference (tests/golden/field_*.json "params").  Only 64-bit words are supported: the MI355X kernels
use u64 limbs with u128 column accumulators (SURVEY 8 sizes).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

WL = 64

# named primes of the hot-path configs (pseudo.py:1498-1548, monty.py:1966-2062) + a few neighbours
NAMED = {
    "X25519": (2**255 - 19, "pseudo"),
    "ED25519": (2**255 - 19, "pseudo"),
    "NIST256": (2**256 - 2**224 + 2**192 + 2**96 - 1, "monty"),
    "X448": (2**448 - 2**224 - 1, "monty"),
    "ED448": (2**448 - 2**224 - 1, "monty"),
    "NIST384": (2**384 - 2**128 - 2**96 + 2**32 - 1, "monty"),
    "NIST224": (2**224 - 2**96 + 1, "monty"),
    "PM266": (2**266 - 3, "pseudo"),
    "C2065": (2**206 - 5, "pseudo"),
    "PM336": (2**336 - 3, "pseudo"),
    "PM383": (2**383 - 187, "pseudo"),
    "NUMS256W": (2**256 - 189, "pseudo"),
    "NIST521": (2**521 - 1, "pseudo"),
    # secp256k1: curve.py:190-198 takes pseudo.py's field at 64 bits (the split-high-part "overflow" form,
    # pseudo.py:1640-1657); monty.py's flavour of the same prime is kept under the key SECP256K1M
    "SECP256K1": (2**256 - 2**32 - 977, "pseudo"),
    "SECP256K1M": (2**256 - 2**32 - 977, "monty"),
    "C41417": (2**414 - 17, "pseudo"),
    # the same pseudo-Mersennes as monty.py builds them when asked (monty.py:2010-2042 names them too): PM383 / PM266 / PM336 in
    # its "exploitable pseudo-Mersenne" form (PM = True, monty.py:700-870), C41417 / PM512 as ordinary full-Montgomery primes
    "PM383M": (2**383 - 187, "monty"),
    "PM266M": (2**266 - 3, "monty"),
    "PM336M": (2**336 - 3, "monty"),
    "C41417M": (2**414 - 17, "monty"),
    "PM512M": (2**512 - 569, "monty"),
    # a 607-bit Mersenne prime: the only kind of modulus that takes pseudo.py's bad_overflow forms at 64 bits (601-610 bits)
    "M607": (2**607 - 1, "pseudo"),
    # the fields of curve.py's ED248 / ED376 / ED500 (monty.py:2095-2102)
    "ED248": (5 * 2**248 - 1, "monty"),
    "ED376": (65 * 2**376 - 1, "monty"),
    "ED500": (27 * 2**500 - 1, "monty"),
    # further named moduli of monty.py's list (monty.py:1990-2075)
    "GM270": (2**270 - 2**162 - 1, "monty"),
    "GM240": (2**240 - 2**183 - 1, "monty"),
    "GM360": (2**360 - 2**171 - 1, "monty"),
    "GM480": (2**480 - 2**240 - 1, "monty"),
    "GM384": (2**384 - 2**186 - 1, "monty"),
    "GM512": (2**512 - 2**127 - 1, "monty"),
    "GM378": (2**378 - 2**324 - 1, "monty"),
    "PM512": (2**512 - 569, "pseudo"),
    "TWEEDLE": (0x40000000000000000000000000000000038aa127696286c9842cafd400000001, "monty"),
    "SIDH434": (2**216 * 3**137 - 1, "monty"),
    "SIDH503": (2**250 * 3**159 - 1, "monty"),
    # the larger isogeny / MFP moduli of the same list (monty.py:2067-2105), round 3
    "SIDH610": (2**305 * 3**192 - 1, "monty"),
    "SIDH751": (2**372 * 3**239 - 1, "monty"),
    "MFP4": (3 * 67 * 2**246 - 1, "monty"),
    "MFP7": (2**145 * 3**9 * 59**3 * 311**3 * 317**3 * 503**3 - 1, "monty"),
    "MFP1973": (0x34e29e286b95d98c33a6a86587407437252c9e49355147ffffffffffffffffff, "monty"),
    "CSIDH512": (5326738796327623094747867617954605554069371494832722337612446642054009560026576537626892113026381253624626941643949444792662881241621373288942880288065659, "monty"),
    # group orders (curve.py:324-329 runs monty.py on "00<decimal q>"): general primes, full Montgomery
    "NIST256Q": (0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551, "monty"),
    "ED25519Q": (0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED, "monty"),
    "ED448Q": ((2**448 - 2**224 - 1 + 1 - 28312320572429821613362531907042076847709625476988141958474579766324) // 4, "monty"),
}

# per-name radix choices the generators hard-wire for 64-bit words (monty.py:2002-2037, `if WL==64: base=...`)
RADIX_64 = {"GM240": 61, "GM360": 57, "GM480": 60, "GM384": 62, "GM512": 58, "MFP4": 52, "MFP7": 52, "MFP1973": 52, "PM512M": 58}

# keys of NAMED that are not the generators' own spelling
REFERENCE_NAME = {"SECP256K1M": "SECP256K1", "PM383M": "PM383", "PM266M": "PM266", "PM336M": "PM336", "C41417M": "C41417", "PM512M": "PM512",
                  "M607": "2**607-1"}

# how each built name is spelled on the reference generators' command line (group orders: "00" + decimal)
def reference_argv(name: str):
    p, fam = NAMED[name]
    script = "pseudo.py" if fam == "pseudo" else "monty.py"
    if name.endswith("Q"):
        return script, "00" + str(p)
    return script, REFERENCE_NAME.get(name, name)


@dataclass
class FieldParams:
    name: str
    family: str                 # "pseudo" | "monty"
    p: int
    n: int                      # bit length of p                      (Nbits)
    radix: int                  # limb width in bits                   (Radix)
    nlimbs: int                 # number of u64 limbs                  (Nlimbs)
    xcess: int                  # nlimbs*radix - n
    nbytes: int                 # ceil(n/8)                            (Nbytes)
    pm1d2: int                  # 2-adicity of p-1
    pe: int                     # progenitor exponent (p-1-e)/(2e), e = 2^pm1d2
    roi: List[int]              # non-trivial root of unity, limbs (plain integer form)
    # pseudo-Mersenne only
    m: int = 0                  # 2^n - p
    mm: int = 0                 # m * 2^xcess  (fold multiplier)
    tw: int = 0                 # top word of p in limb nlimbs-1 (as added by caddp)
    overflow: bool = False
    bad_overflow: bool = False  # pseudo.py:1646-1648: the carried high part of the overflow form needs a double word
    fred: bool = False
    epm: bool = False
    carry_on: bool = False
    # Montgomery only
    ppw: List[int] = field(default_factory=list)   # signed prime limbs (+ virtual limb if E)
    E: bool = False
    R: int = 1
    ndash: int = 1
    r2: List[int] = field(default_factory=list)    # R^2 mod p, limbs (nres multiplier)
    trin: int = 0
    pm: bool = False            # monty.py's PM form: an exploitable pseudo-Mersenne given to monty.py (ppw[0] = -m)
    # derived for both families: non-zero prime limbs as (index, sign, magnitude) with the virtual
    # limb folded into limb nlimbs-1 as +2^radix (caddp/addp/subp: pseudo.py:202-220, monty.py:301-349)
    pp: List[tuple] = field(default_factory=list)

    @property
    def montgomery(self) -> bool:
        return self.family == "monty"

    def to_limbs(self, x: int, masked_top: bool = False) -> List[int]:
        """integer -> limbs; top limb takes everything left unless masked_top (pseudo.py:1769-1775)."""
        b = 1 << self.radix
        out = []
        for _ in range(self.nlimbs - 1):
            out.append(x % b)
            x >>= self.radix
        out.append(x % b if masked_top else x)
        return out

    def from_limbs(self, limbs) -> int:
        return sum(int(v) << (self.radix * i) for i, v in enumerate(limbs))


def _bits(x: int) -> int:
    return x.bit_length()


def _makebig(x: int, radix: int, n: int) -> List[int]:
    return [(x >> (radix * i)) & ((1 << radix) - 1) for i in range(n)]


def _two_adic(p: int):
    q, k = p - 1, 0
    while q % 2 == 0:
        q //= 2
        k += 1
    e = 1 << k
    return k, (p - 1 - e) // (2 * e)


def _root_of_unity(p: int, k: int) -> int:
    if k == 1:
        return p - 1
    if k == 2:
        return pow(2, (p - 1) // 4, p)
    qnr = 2
    while pow(qnr, (p - 1) // 2, p) == 1:
        qnr += 1
    return pow(qnr, (p - 1) >> k, p)


# ------------------------------------------------------------------ pseudo-Mersenne (2^n - m)
def _pm_radix(n: int) -> int:
    """smallest limb count whose worst-case column sum fits 2*WL bits, radix <= WL-3
    (rule of pseudo.py:124-140)."""
    limbs = n // WL
    while True:
        limbs = max(limbs + 1, 2)
        radix = WL // 2
        while limbs * radix < n:
            radix += 1
        if radix > WL - 3:
            continue
        if limbs * ((1 << radix) - 1) ** 2 < 1 << (2 * WL):
            return radix


def derive_pseudo(name: str, p: int, radix: Optional[int] = None) -> FieldParams:
    n = p.bit_length()
    if n < 120 or pow(3, p - 1, p) != 1:
        raise ValueError("not a sensible modulus")
    radix = radix or _pm_radix(n)
    m = (1 << n) - p
    b = 1 << radix
    if m >= b:
        raise ValueError("not an exploitable pseudo-Mersenne; use the Montgomery family")
    N = -(-n // radix)
    xcess = N * radix - n
    mm = m << xcess
    if mm >= 1 << (WL - 1):
        raise ValueError("excess too large for this radix")
    tw = b if n % radix == 0 else 1 << (n % radix)
    overflow = (b - 1) * (b - 1) * mm * N >= 1 << (2 * WL)
    bad_overflow = overflow and (N - 1) * (b - 1) ** 2 >= 1 << (2 * WL - 3)      # pseudo.py:1646-1648 (no Karatsuba at 64 bits)
    fred = _bits(N + 1) + radix + _bits(mm) < WL
    epm = (not overflow) and mm * (b - 1) < 1 << WL
    carry_on = m * ((1 << (2 * WL - radix + xcess)) + (1 << (radix - xcess))) >= 1 << (2 * radix)
    k, pe = _two_adic(p)
    fp = FieldParams(name=name, family="pseudo", p=p, n=n, radix=radix, nlimbs=N, xcess=xcess,
                     nbytes=-(-n // 8), pm1d2=k, pe=pe, roi=_makebig(_root_of_unity(p, k), radix, N),
                     m=m, mm=mm, tw=tw, overflow=overflow, bad_overflow=bad_overflow, fred=fred, epm=epm, carry_on=carry_on)
    fp.pp = [(0, -1, m), (N - 1, +1, tw)]
    return fp


# ------------------------------------------------------------------ Montgomery, shape-aware
def _signed_limbs(p: int, radix: int, N: int, pm_m: int = 0):
    """rewrite limbs equal to 2^radix-1 as -1 with a carry into the next limb; a carry out of the
    top limb becomes a virtual extra limb (process_prime, monty.py:258-298).  pm_m = m for the PM form: a low limb
    2^radix - m becomes -m, with the same carry (monty.py:284-288)."""
    b = 1 << radix
    pw = _makebig(p, radix, N)
    out, carry = [], 0
    for i in range(N):
        v = pw[i] + carry
        if pm_m and i == 0:
            if v == b - pm_m:
                v, carry = -pm_m, 1
            out.append(v)
            continue
        if carry:
            if v == b - 1:
                v = -1                      # carry stays 1
            elif v == b:
                v, carry = 0, 1
            else:
                carry = 0
        elif v == b - 1:
            v, carry = -1, 1
        out.append(v)
    if carry:
        out.append(1)
    return out, bool(carry)


def _monty_radix(p: int, n: int) -> int:
    """default radix rule of monty.py:151-173: at least two spare bits in the top limb, or none at
    all together with a virtual limb."""
    limbs = n // WL
    while True:
        limbs = max(limbs + 1, 2)
        radix = WL // 2

        def bump(r):
            while limbs * r < n or r - (n % r) < 2:
                r += 1
            return r
        radix = bump(radix)
        if n % radix == 0:
            _, E = _signed_limbs(p, radix, limbs)
            if not E:
                radix += 1
        radix = bump(radix)
        if radix > WL - 3:
            continue
        if limbs * ((1 << radix) - 1) ** 2 < 1 << (2 * WL):
            return radix


def _trinomial(p: int, radix: int) -> int:
    """p = 2^n - 2^k - 1 with k a multiple of the radix -> k/radix, else 0 (monty.py:231-243)."""
    n = p.bit_length()
    m = (1 << n) - p - 1
    k = 20
    while k < n:
        if 1 << k > m:
            return 0
        if 1 << k == m:
            break
        k += 1
    else:
        return 0
    return k // radix if k % radix == 0 else 0


def derive_monty(name: str, p: int, radix: Optional[int] = None) -> FieldParams:
    n = p.bit_length()
    if n < 120 or pow(3, p - 1, p) != 1:
        raise ValueError("not a sensible modulus")
    radix = radix or _monty_radix(p, n)
    b = 1 << radix
    N = -(-n // radix)
    xcess = N * radix - n
    m = (1 << n) - p
    pm = m > 1 and _bits(m) + radix < WL               # "Exploitable Pseudo Mersenne detected" (monty.py:2151-2156)
    ppw, E = _signed_limbs(p, radix, N, m if pm else 0)
    if sum(1 for i, v in enumerate(ppw) if i > 0 and v == -1) > 1:
        raise ValueError("too many -1 limbs (monty.py:2217-2219)")
    if xcess < 2 and not E:
        raise ValueError("excess is only one bit; change the radix")
    R = 1 << (radix * (N + (1 if E else 0)))
    ndash = pow((R - p) % b, -1, b)
    k, pe = _two_adic(p)
    fp = FieldParams(name=name, family="monty", p=p, n=n, radix=radix, nlimbs=N, xcess=xcess,
                     nbytes=-(-n // 8), pm1d2=k, pe=pe, roi=_makebig(_root_of_unity(p, k), radix, N),
                     m=m, ppw=ppw, E=E, R=R, ndash=ndash, r2=_makebig(R * R % p, radix, N),
                     trin=_trinomial(p, radix), pm=pm)
    pp = [(i, -1 if v < 0 else +1, abs(v)) for i, v in enumerate(ppw[:N]) if v]
    if E:
        # fold +1 * 2^(radix*N) into limb N-1 as +2^radix
        # (a negative top limb, GM378's -1 at limb N-1, joins it: -x + x*q = x*(q-1) in the 64-bit arithmetic of
        # caddp / addp / subp, monty.py:301-349)
        pp = [t for t in pp if t[0] != N - 1] + [(N - 1, +1, ppw[N - 1] + b)]
    fp.pp = pp
    return fp


def derive(name: str, family: Optional[str] = None, radix: Optional[int] = None) -> FieldParams:
    """FieldParams for a named prime (or a python expression such as '2**255-19')."""
    if name in NAMED:
        p, fam = NAMED[name]
    else:
        p, fam = int(eval(name, {"__builtins__": {}})), None
    fam = family or fam
    if fam is None:
        n = p.bit_length()
        fam = "pseudo" if ((1 << n) - p) < (1 << 32) else "monty"
    if radix is None and fam == "monty":
        radix = RADIX_64.get(name)
    return derive_pseudo(name, p, radix) if fam == "pseudo" else derive_monty(name, p, radix)
