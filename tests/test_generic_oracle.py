"""The run-time generic oracle (oracle/field_generic.c) against the reference-generated golden vectors of
EVERY built prime -- the three BASELINE primes (where it must also agree with the per-prime restatements) and
the further ones, for which it is the CPU oracle.  CPU only."""
import pytest

from tests.conftest import limbs, load_golden
from tests.generic_oracle import Generic

from modarith_amd.emit import BUILT_PRIMES

from modarith_amd import generate as _gen

ALL = list(BUILT_PRIMES) + [_gen.resolve(arg, fam).name for arg, fam in _gen.EXAMPLES]     # + the generator mode's unnamed moduli


@pytest.fixture(scope="module", params=ALL)
def gx(request, oracle):
    P = request.param
    return P, Generic(oracle.lib, P), load_golden("field_%s.json" % P)


def test_all_limb_exact_ops(gx):
    P, G, g = gx
    o = g["ops"]
    for i, (a, b) in enumerate(zip(g["A"], g["B"])):
        a, b = limbs(a), limbs(b)
        for op in ("modadd", "modsub", "modmul"):
            assert G.bi(op, a, b) == limbs(o[op][i]), (P, op, i)
        for op in ("modneg", "modsqr", "redc", "nres"):
            assert G.un(op, a) == limbs(o[op][i]), (P, op, i)
        C, D = G.bi("modmul", a, b), G.bi("modsub", a, b)
        assert G.bi("modmul", C, D) == limbs(o["chain_mul_CD"][i])
        assert G.un("modsqr", D) == limbs(o["chain_sqr_D"][i])
        x = G.arr(a)
        r = G.lib.gen_modfsb(G.R, x)
        assert list(x) == limbs(o["modfsb"][i][0]) and int(r) == o["modfsb"][i][1]
    for a, row in zip(g["A"], o["modmli"]):
        for k, want in zip(o["modmli_ints"], row):
            z = G.arr()
            G.lib.gen_modmli(G.R, G.arr(limbs(a)), k, z)
            assert list(z) == limbs(want), (P, "modmli", k)


def test_inverse_sqrt_after_redc(gx):
    P, G, g = gx
    for rec in g["ops"]["modinv"][:12]:
        z = G.arr()
        G.lib.gen_modinv(G.R, G.arr(limbs(rec["x"])), None, z)
        assert G.un("redc", list(z)) == limbs(rec["inv_redc"])
    for rec in load_golden("sqrt_%s.json" % P)["recs"][:12]:
        r = G.arr()
        G.lib.gen_modsqrt(G.R, G.arr(limbs(rec["x"])), None, r)
        assert G.un("redc", list(r)) == limbs(rec["sqrt_redc"])
        assert G.lib.gen_modqr(G.R, None, G.arr(limbs(rec["x"]))) == rec["qr"]


@pytest.mark.parametrize("P", ["X25519", "NIST256", "X448"])
def test_agrees_with_per_prime_restatement(oracle, P):
    import random
    from modarith_amd.params import derive
    G = Generic(oracle.lib, P)
    fp = derive(P)
    rng = random.Random(5)
    for _ in range(300):
        a, b = fp.to_limbs(rng.randrange(0, 2 * fp.p)), fp.to_limbs(rng.randrange(0, 2 * fp.p))
        for op in ("modmul", "modadd", "modsub"):
            assert G.bi(op, a, b) == oracle.bi(op, P, a, b)
        assert G.un("modsqr", a) == oracle.un("modsqr", P, a)


def test_round2_modnsqr_and_out_of_contract(gx):
    """tests/golden/make_golden_r2.py: the reference's modnsqr (k squarings in place, pseudo.py:745-755) and its
    answers on out-of-contract limbs (64-bit wrap-around), for every built prime"""
    P, G, _ = gx
    g2 = load_golden("field_%s_r2.json" % P)
    for rec in g2["modnsqr"]:
        z = limbs(rec["a"])
        for _ in range(rec["k"]):
            z = G.un("modsqr", z)
        assert z == limbs(rec["out"]), (P, "modnsqr", rec["k"])
    for i, rec in enumerate(g2["ooc"]):
        a, b = limbs(rec["a"]), limbs(rec["b"])
        for op in ("modmul", "modadd", "modsub"):
            assert G.bi(op, a, b) == limbs(rec[op]), (P, op, i)
        for op in ("modsqr", "nres", "redc", "modneg"):
            assert G.un(op, a) == limbs(rec[op]), (P, op, i)
        z = G.arr()
        G.lib.gen_modmli(G.R, G.arr(a), 121665, z)
        assert list(z) == limbs(rec["modmli_121665"]), (P, "modmli", i)
