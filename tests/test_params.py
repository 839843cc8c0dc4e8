"""The parameter driver (modarith_amd/params.py, emit.py) against the constants captured from the
reference generators (tests/golden/field_*.json "params").  CPU only."""
import os
import random

import pytest

from modarith_amd import emit
from modarith_amd import generate as gen
from modarith_amd.params import derive as _derive_named
from tests.conftest import load_golden

# the built-in primes, and the unnamed moduli of the generator mode (modarith_amd.generate.EXAMPLES) under their tags
GENERATED = {gen.resolve(arg, fam).name: (arg, fam) for arg, fam in gen.EXAMPLES}
ALL = list(emit.BUILT_PRIMES) + list(GENERATED)


def derive(P):
    return gen.resolve(*GENERATED[P]) if P in GENERATED else _derive_named(P)


def _i(v):
    return int(v, 16) if isinstance(v, str) else int(v)


@pytest.mark.parametrize("P", ALL)
def test_driver_matches_reference_constants(P):
    g = load_golden("field_%s.json" % P)["params"]
    fp = derive(P)
    assert (fp.n, fp.radix, fp.nlimbs, fp.xcess, fp.nbytes, fp.pm1d2) == (g["n"], g["base"], g["N"], g["xcess"], g["Nbytes"], g["PM1D2"])
    assert fp.p == _i(g["p"]) and fp.pe == _i(g["PE"])
    assert fp.roi == [_i(v) for v in g["ROI"]]
    hdr = " ".join(g["header"])
    for key, val in (("Nlimbs", fp.nlimbs), ("Radix", fp.radix), ("Nbits", fp.n), ("Nbytes", fp.nbytes)):
        assert "#define %s %d" % (key, val) in hdr
    assert ("#define MONTGOMERY" in hdr) == fp.montgomery
    if fp.family == "pseudo":
        assert (fp.m, fp.mm, fp.tw) == (_i(g["m"]), _i(g["mm"]), _i(g["TW"]))
        assert (fp.overflow, fp.fred, fp.epm, fp.carry_on) == (g["overflow"], g["fred"], g["EPM"], g["carry_on"])
        assert fp.bad_overflow == g["bad_overflow_mul"] == g["bad_overflow_sqr"]
    else:
        assert fp.ppw == [(-_i(v[1:]) if v.startswith("-") else _i(v)) for v in g["ppw"]]
        assert (fp.E, fp.R, fp.ndash, fp.trin) == (g["E"], _i(g["R"]), _i(g["ndash"]), g["trin"])
        assert fp.r2 == [_i(v) for v in g["cw"]]
        assert g["fullmonty"] is (fp.ndash != 1) and g["PM"] is fp.pm
        if fp.pm:
            assert _i(g["M"]) == fp.m == -fp.ppw[0]


def test_reference_stdout_lines():
    """the generators' own log lines (SURVEY 8(a) "pinned by")"""
    log = "\n".join(load_golden("field_X25519.json")["params"]["log"])
    assert "Chosen radix is 51 bits, using 5 limbs with excess of 0 bits" in log
    assert "Tighter reduction" in log and "Fully Exploitable Pseudo-Mersenne detected" in log
    log = "\n".join(load_golden("field_X448.json")["params"]["log"])
    assert "Extra virtual limb added" in log and "lucky trinomial" in log


@pytest.mark.parametrize("P", ALL)
def test_addition_chain_computes_progenitor(P):
    fp = derive(P)
    prog = emit.addition_chain(fp.pe)
    rng = random.Random(1)
    for _ in range(4):
        x = rng.randrange(2, fp.p)
        assert emit.eval_chain(prog, x, fp.p) == pow(x, fp.pe, fp.p)
    sq, mu = emit.chain_cost(prog)
    assert sq <= fp.pe.bit_length()               # squarings == bit length - 1: the leading run ladder is the main chain
    if not P.endswith("Q") and P not in ("BP256", "TWEEDLE", "SIDH434", "SIDH503", "SIDH610", "SIDH751", "MFP4", "MFP7", "MFP1973", "CSIDH512", "M607"):
        assert mu <= 20                           # shaped primes: long runs of ones; general primes take the loop form


def test_generated_headers_are_current():
    """csrc/generated/params_*.h in the tree equal what the driver emits now"""
    for P in emit.BUILT_PRIMES:
        path = os.path.join(emit.GEN_DIR, "params_%s.h" % P)
        assert os.path.exists(path), "run python -m modarith_amd.emit"
        assert open(path).read() == emit.header_text(derive(P))


def test_limb_split_roundtrip():
    for P in ALL:
        fp = derive(P)
        for x in (0, 1, fp.p - 1, fp.p, 2 * fp.p - 1):
            assert fp.from_limbs(fp.to_limbs(x)) == x


def test_split_proofs_per_prime():
    """emit.split_point / chain_ok: the three-accumulator cut positions the kernels were tuned with stay what they were, and the two
    primes that had no provable cut under the dense count of 2N column terms (ED500, SIDH503: Montgomery primes with ndash = 1 whose
    limbs are mostly 0 / -1 / powers of two, which never enter the accumulators -- field.h monty_reduce) get one from the per-prime
    count (emit.sparse_terms); a bound check of that count against the worst column redone here"""
    from modarith_amd import emit
    from modarith_amd.params import derive
    want = {"X25519": (28, True), "NIST256": (27, True), "X448": (29, True), "NIST384": (29, True), "SECP256K1": (27, False), "ED248": (26, True),
            "ED376": (28, True), "NIST521": (0, False), "ED500": (29, False), "SIDH503": (29, False), "SIDH751": (0, False), "CSIDH512": (0, False)}
    for name, (h, chain) in want.items():
        fp = derive(name)
        assert (emit.split_point(fp), emit.chain_ok(fp)) == (h, chain), name
    for name in ("ED500", "SIDH503"):
        fp = derive(name)
        n, H, W = emit.sparse_terms(fp), emit.split_point(fp), fp.radix + 2
        big = [v for i, v in enumerate(fp.ppw) if i > 0 and v not in (0, 1, -1) and v & (v - 1)]
        assert n == fp.nlimbs + len(big) + 1 and 2 * fp.nlimbs > n            # sparse indeed: fewer terms than the dense count
        lo, hi = (1 << H) - 1, (1 << (W - H)) - 1                              # halves of a limb below 2^W
        assert n * lo * lo < 1 << 64 and n * 2 * lo * hi < 1 << 64 and n * hi * hi < 1 << 64
        assert all(0 < v < 1 << fp.radix for v in big)                         # the prime limbs themselves are narrower than a limb
    assert emit.sparse_terms(derive("NIST384")) is None                        # ndash != 1: the dense count stands
