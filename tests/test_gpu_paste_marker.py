"""The paste-marker consumer on the GPU: examples/paste_marker_consumer.c -- RFC 7748 section 5 written against the undecorated names and
macros of a pasted field.c, nothing else -- includes include/field_<PRIME>.h where the reference says "paste field.c here", is compiled
with gcc, linked with libmodarith_amd.so, and must print the RFC 7748 public key and the value of the reference's chained-call loop
(rfc7748.c:297-305: LCG-keyed scalar, (k, u) -> v, (k, v) -> u) at the survey-captured checkpoints (tests/golden/ladder_*.json
ref_main_chain).  Every field call is one element through the device (about 4 700 calls per scalar multiplication), so the default run
checks the checkpoint after 100 steps (X25519, 200 scalar multiplications, ~50 s) / 10 steps (X448); MA_FULL_CHAIN=1 runs the 5 000 steps of
the reference's main() (34 minutes: profiles/r06_paste_marker_chain_5000.log, final value 2ac5ee20...b1d41c as the reference prints it;
profiles/r06_paste_marker_chain.log is a 1 000-step run)."""
import os
import re
import subprocess

import pytest

from tests.conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve,flag,steps", [("X25519", "", 100), ("X448", "-DUSE_X448", 10)])
def test_consumer_written_against_the_pasted_names_reproduces_the_reference(curve, flag, steps, tmp_path):
    if os.environ.get("MA_FULL_CHAIN") == "1":
        steps = 5000
    g = load_golden("ladder_%s.json" % curve)
    exe = str(tmp_path / "consumer")
    cmd = ["gcc", "-O2", os.path.join(ROOT, "examples", "paste_marker_consumer.c"), "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "modarith_amd"),
           "-l:libmodarith_amd.so", "-Wl,-rpath," + os.path.join(ROOT, "modarith_amd"), "-o", exe] + ([flag] if flag else [])
    subprocess.run(cmd, check=True, timeout=300)
    p = subprocess.run([exe, str(steps)], capture_output=True, text=True, timeout=3000 if steps <= 100 else 40000)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-1000:]
    out = dict(l.split(" ", 1) for l in p.stdout.strip().splitlines())
    nl, radix, nbits, nbytes = {"X25519": (5, 51, 255, 32), "X448": (8, 56, 448, 56)}[curve]
    assert out["field"] == "Wordlength 64 Nlimbs %d Radix %d Nbits %d Nbytes %d sizeof(spint) 8" % (nl, radix, nbits, nbytes)
    assert out["vector"] == g["kat"][0]["out"]                                      # RFC 7748 6.1 / 6.2
    assert out["key"] == g["ref_main_chain"]["bk"] and out["steps"] == str(steps)
    assert out["chain"] == g["ref_main_chain"]["checkpoints"][str(steps)]


def test_prop_entry_point_returns_the_references_mask():
    """prop is static in field.c (pseudo.py:223-251); the library exports it so that all 32 emitted names resolve.  Against a
    big-integer model of the emitted code on limbs with signed excess, and its batched form on the same rows"""
    import ctypes
    import random
    import torch
    from modarith_amd import _lib
    from modarith_amd.field import Field
    L = _lib.load()
    rng = random.Random(5)
    for P, (nl, radix) in (("X25519", (5, 51)), ("NIST256", (5, 52)), ("X448", (8, 56))):
        fn = getattr(L, "prop_%s_ct" % P)
        fn.argtypes, fn.restype = [ctypes.POINTER(ctypes.c_uint64)], ctypes.c_uint64
        mask = (1 << radix) - 1
        rows, want, wmask = [], [], []
        for it in range(40):
            n = [rng.getrandbits(64) if it % 2 else (rng.getrandbits(radix + 3) - (rng.getrandbits(radix) if it % 4 == 0 else 0)) % 2**64 for _ in range(nl)]
            s = lambda v: v - 2**64 if v >= 2**63 else v
            m = list(n)
            carry = s(m[0]) >> radix
            m[0] &= mask
            for i in range(1, nl - 1):
                carry += s(m[i])
                m[i] = carry & mask
                carry >>= radix
            m[nl - 1] = (m[nl - 1] + carry) % 2**64
            rows.append(n); want.append(m); wmask.append(2**64 - 1 if m[nl - 1] >> 63 else 0)
            buf = (ctypes.c_uint64 * nl)(*n)
            r = fn(buf)
            assert list(buf) == m and r == wmask[-1], (P, it)
        F = Field(P)
        t = F.from_limbs(rows)
        flag = F.prop(t)
        assert F.to_limbs(t) == want
        assert flag.tolist() == [-1 if w else 0 for w in wmask]


ORACLE_PROTOS = """
spint flatten_%(P)s(spint *); spint modfsb_%(P)s(spint *);
void modadd_%(P)s(const spint *, const spint *, spint *); void modsub_%(P)s(const spint *, const spint *, spint *); void modneg_%(P)s(const spint *, spint *);
void modmli_%(P)s(const spint *, int, spint *); void modmul_%(P)s(const spint *, const spint *, spint *); void modsqr_%(P)s(const spint *, spint *);
void modcpy_%(P)s(const spint *, spint *); void modnsqr_%(P)s(spint *, int); void modpro_%(P)s(const spint *, spint *);
void modinv_%(P)s(const spint *, const spint *, spint *); void nres_%(P)s(const spint *, spint *); void redc_%(P)s(const spint *, spint *);
int modis1_%(P)s(const spint *); int modis0_%(P)s(const spint *); void modzer_%(P)s(spint *); void modone_%(P)s(spint *); void modint_%(P)s(int, spint *);
int modqr_%(P)s(const spint *, const spint *); void modcmv_%(P)s(int, const spint *, volatile spint *); void modcsw_%(P)s(int, volatile spint *, volatile spint *);
void modsqrt_%(P)s(const spint *, const spint *, spint *); void modshl_%(P)s(unsigned int, spint *); int modshr_%(P)s(unsigned int, spint *);
void modhaf_%(P)s(spint *); void mod2r_%(P)s(unsigned int, spint *); void modexp_%(P)s(const spint *, char *); int modimp_%(P)s(const char *, spint *);
int modsign_%(P)s(const spint *); int modcmp_%(P)s(const spint *, const spint *);
/* prop is static in field.c and in the oracle alike: the emitted form (pseudo.py:223-251, arithmetic shift) */
static spint prop(spint *n) {
    spint mask = ((spint)1 << Radix) - (spint)1;
    sspint carry = (sspint)n[0];
    carry >>= Radix;
    n[0] &= mask;
    for (int i = 1; i < Nlimbs - 1; i++) { carry += (sspint)n[i]; n[i] = (spint)carry & mask; carry >>= Radix; }
    n[Nlimbs - 1] += (spint)carry;
    return -((n[Nlimbs - 1] >> 1) >> (Wordlength - 2));
}
"""


@pytest.mark.parametrize("P", ["X25519", "NIST256", "X448"])
def test_all_32_names_through_the_pasted_header_equal_the_oracle(P, tmp_path):
    """examples/paste_marker_all32.c calls every function of field.c through the undecorated names.  Built against include/field_<P>.h
    + libmodarith_amd.so it runs on the GPU; built against a header that maps the same names onto oracle/liboracle.so it runs the CPU
    restatement of the reference's emitted code.  Same program, same inputs: the outputs must be equal line for line -- limbs of the
    add / sub / product / shift / conversion functions exactly, modpro / modinv / modsqrt by value."""
    from tests.oracle_binding import build_oracle
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        build_oracle()
    shim = open(os.path.join(ROOT, "include", "field_%s.h" % P)).read()
    macros = [l for l in shim.splitlines() if l.startswith("#define ") and "_ct" not in l and "_H" not in l]
    names = [l.split()[1] for l in shim.splitlines() if l.startswith("#define ") and l.endswith("_%s_ct" % P)]
    assert len(names) == 32
    ohdr = tmp_path / "oracle_field.h"
    ohdr.write_text("#include <stdio.h>\n#include <stdint.h>\n" + "\n".join(macros) + "\n" + ORACLE_PROTOS % {"P": P}
                    + "\n".join("#define %s %s_%s" % (n, n, P) for n in names if n != "prop") + "\n")
    src = os.path.join(ROOT, "examples", "paste_marker_all32.c")
    outs = {}
    for tag, hdr, incs, libs in (("gpu", '"field_%s.h"' % P, ["-I" + os.path.join(ROOT, "include")],
                                  ["-L" + os.path.join(ROOT, "modarith_amd"), "-l:libmodarith_amd.so", "-Wl,-rpath," + os.path.join(ROOT, "modarith_amd")]),
                                 ("oracle", '"oracle_field.h"', ["-I" + str(tmp_path)],
                                  ["-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])):
        exe = str(tmp_path / ("all32_" + tag))
        subprocess.run(["gcc", "-O2", "-DFIELD_HEADER=" + hdr, src] + incs + libs + ["-o", exe], check=True, timeout=300)
        p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (tag, p.stdout[-500:], p.stderr[-500:])
        outs[tag] = p.stdout.strip().splitlines()
    assert len(outs["gpu"]) >= 40 and [l.split()[0] for l in outs["gpu"]] == [l.split()[0] for l in outs["oracle"]]
    for g, o in zip(outs["gpu"], outs["oracle"]):
        assert g == o, (P, g, o)
    got = dict((l.split()[0], l.split()[1:]) for l in outs["gpu"])
    assert got["inv*a==1"] == ["1"] and got["sqrt^2==x"] == ["1"] and got["modimp"] == ["3"] and got["modcmp"] == ["1"]
