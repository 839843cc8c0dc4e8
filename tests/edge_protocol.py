"""The reference's corner-case protocol (edge.py:116-163 vector generation, 341-360 the 17 corner pairs; edge.c:77-247 the
checks), restated for the parity tests: every pair (a, b) is imported with modimp and must satisfy, through modcmp,

    1/(1/a) = a,  a+b,  a-b,  b-a,  a*b,  sqr(sqrt(sqr(a))) = a^2

and then -- after `modadd(x,x,x); modadd(y,y,y)` ("double them and try again", edge.c:167) -- the same six identities on the
doubled operands 2a, 2b.  The expected values are plain integers mod p; comparison is modcmp (redc both sides), as in
edge.c, so the addition chain of modpro does not matter.  `engine` supplies the field functions: the CPU oracle (one element
at a time) or the HIP library (all 17 pairs as one batch)."""
import random


def corner_pairs(p: int, n: int, seed: int = 42):
    """edge.py:341-358 (r random below p, i its inverse, c = 2^n - p; every entry positive and below 2^n)"""
    rng = random.Random(seed)
    r = rng.randrange(0, p)
    i = pow(r, p - 2, p)
    c = (1 << n) - p
    return [(p - 1, p - 1), (0, p - 1), (0, 0), (r, r), (r, i), (p, 1), (p - 1, 1), (p - 2, 2), (p - 1, r), (p - 2, r), (1 << 64, 1 << 64),
            (1 << (n - 1), (1 << (n - 1)) - 1), (c, 1), (c, (1 << n) - 1), ((1 << n) - 1, 0), ((1 << n) - 1, 1), ((1 << n) - 1, (1 << n) - 1)]


def expected(p: int, a: int, b: int):
    """edge.py:116-163: the twelve check values of one pair, in file order"""
    return [a % p, (a + b) % p, (a - b) % p, (b - a) % p, (a * b) % p, (a * a) % p,
            (2 * a) % p, (2 * a + 2 * b) % p, (2 * a - 2 * b) % p, (2 * b - 2 * a) % p, (2 * a * 2 * b) % p, (2 * a * 2 * a) % p]


CHECKS = ("1/(1/a)", "modadd(a,b)", "modsub(a,b)", "modsub(b,a)", "modmul(a,b)", "modsqr(modsqrt(modsqr(a)))",
          "1/(1/2a)", "modadd(2a,2b)", "modsub(2a,2b)", "modsub(2b,2a)", "modmul(2a,2b)", "modsqr(modsqrt(modsqr(2a)))")


def run(engine, p: int, n: int, nbytes: int, seed: int = 42):
    """returns the list of (pair index, check name) that FAILED; engine methods work on opaque batches:
    imp(list of ints) -> batch, inv, add, sub, mul, sqr, sqrt, cmp(batch, batch) -> list of 0/1"""
    pairs = corner_pairs(p, n, seed)
    want = [expected(p, a, b) for a, b in pairs]
    W = [engine.imp([w[k] for w in want]) for k in range(12)]
    x = engine.imp([a for a, _ in pairs])
    y = engine.imp([b for _, b in pairs])
    fails = []

    def six(x, y, base):
        z = [engine.inv(engine.inv(x)), engine.add(x, y), engine.sub(x, y), engine.sub(y, x), engine.mul(x, y),
             engine.sqr(engine.sqrt(engine.sqr(x)))]
        for k, zz in enumerate(z):
            ok = engine.cmp(W[base + k], zz)
            fails.extend((j, CHECKS[base + k]) for j, v in enumerate(ok) if not v)
    six(x, y, 0)
    x2, y2 = engine.add(x, x), engine.add(y, y)          # edge.c:167
    six(x2, y2, 6)
    return fails
