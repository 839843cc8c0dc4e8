#!/usr/bin/env python3
"""Round-2 additions to the golden fixtures, again produced by RUNNING THE REFERENCE in this container
(same machinery as make_golden.py: refgen.py drives the unmodified generators, gcc compiles the emitted C,
ctypes calls it).  Writes tests/golden/field_<PRIME>_r2.json with

  * "modnsqr": the reference's modnsqr(a, k) (pseudo.py:745-755, monty.py:1182-1192) on contract inputs;
  * "ooc": OUT-OF-CONTRACT limbs -- every limb drawn from {0, 1, 2^R-1, 2^R, 2^(R+1)-1, 2^(R+2)-1, 2^(R+2),
    2^(R+3)-1, 2^63, 2^64-1, random 64-bit, random R-bit} -- through the emitted modmul / modsqr / nres / redc /
    modadd / modsub / modneg / modmli.  field.c signals no error for such inputs; what it returns is defined by its
    64-bit wrap-around, and the engine's exact product path claims to reproduce exactly that.

  python tests/golden/make_golden_r2.py
"""
import json, os, random, sys
from ctypes import c_int

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Ref, element_pool, hx  # noqa: E402
import gio  # noqa: E402


def limb_classes(R, rng):
    M = (1 << 64) - 1        # limbs are 64-bit words (radix 62 puts 2^(R+2) beyond them)
    return [v & M for v in (0, 1, (1 << R) - 1, 1 << R, (1 << (R + 1)) - 1, (1 << (R + 2)) - 1, 1 << (R + 2), (1 << (R + 3)) - 1,
                            1 << 63, (1 << 64) - 1, rng.getrandbits(64), rng.getrandbits(R))]


def fixture(script, arg, seed, name, count=96):
    rng = random.Random(seed)
    ref = Ref(script, arg)
    N, R = ref.N, ref.base
    H = lambda limbs: [hx(v) for v in limbs]
    fx = {"prime": name, "generator": script, "seed": seed, "radix": R, "nlimbs": N}
    # modnsqr on contract inputs (internal form: nres of pool values / raw unmasked-top limbs)
    raw, _ = element_pool(ref, rng, 48)
    A = [ref.un("nres", a) if i % 2 == 0 else a for i, a in enumerate(raw)]
    ks = [0, 1, 2, 3, 5, 10, 50]
    recs = []
    for i, a in enumerate(A):
        k = ks[i % len(ks)]
        z = ref.arr(a)
        ref.lib.modnsqr(z, c_int(k))
        recs.append({"a": H(a), "k": k, "out": H(list(z))})
    fx["modnsqr"] = recs
    # out-of-contract limbs
    ooc = []
    for i in range(count):
        cl = limb_classes(R, rng)
        if i < len(cl):                       # every class once in every limb position at the same time
            a = [cl[i]] * N
            b = [cl[(i * 5 + 3) % len(cl)]] * N
        else:
            a = [rng.choice(limb_classes(R, rng)) for _ in range(N)]
            b = [rng.choice(limb_classes(R, rng)) for _ in range(N)]
        z = ref.arr()
        ref.lib.modmli(ref.arr(a), c_int(121665), z)
        ooc.append({"a": H(a), "b": H(b),
                    "modmul": H(ref.bi("modmul", a, b)), "modsqr": H(ref.un("modsqr", a)),
                    "nres": H(ref.un("nres", a)), "redc": H(ref.un("redc", a)),
                    "modadd": H(ref.bi("modadd", a, b)), "modsub": H(ref.bi("modsub", a, b)),
                    "modneg": H(ref.un("modneg", a)), "modmli_121665": H(list(z))})
    fx["ooc"] = ooc
    return fx


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from modarith_amd.emit import EXTRA_PRIMES
    from modarith_amd.params import reference_argv
    jobs = [("pseudo.py", "X25519", "X25519"), ("monty.py", "NIST256", "NIST256"), ("monty.py", "X448", "X448")]
    jobs += [reference_argv(n) + (n,) for n in EXTRA_PRIMES]
    from modarith_amd.generate import EXAMPLES, resolve            # unnamed moduli of the generator mode (appended: seeds of the others stay)
    jobs += [("pseudo.py" if fam == "pseudo" else "monty.py", arg.split("=", 1)[-1], resolve(arg, fam).name) for arg, fam in EXAMPLES]
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for k, (script, arg, name) in enumerate(jobs):
        if only and name not in only:
            continue
        fx = fixture(script, arg, 9000 + k, name, count=96 if k < 3 else 32)
        gio.dump(fx, "field_%s_r2.json" % name)
        print(name, len(fx["modnsqr"]), "modnsqr records,", len(fx["ooc"]), "out-of-contract records")


if __name__ == "__main__":
    main()
