"""The paste-marker consumer on the GPU: examples/paste_marker_consumer.c -- RFC 7748 section 5 written against the undecorated names and
macros of a pasted field.c, nothing else -- includes include/field_<PRIME>.h where the reference says "paste field.c here", is compiled
with gcc, linked with libmodarith_amd.so, and must print the RFC 7748 public key and the value of the reference's chained-call loop
(rfc7748.c:297-305: LCG-keyed scalar, (k, u) -> v, (k, v) -> u) at the survey-captured checkpoints (tests/golden/ladder_*.json
ref_main_chain).  Every field call is one element through the device (about 4 700 calls per scalar multiplication), so the default run
checks the checkpoint after 100 steps (X25519, 200 scalar multiplications, ~50 s) / 10 steps (X448); MA_FULL_CHAIN=1 runs the 5 000 steps of
the reference's main() (about 40 minutes; profiles/r06_paste_marker_chain.log holds a 1 000-step run)."""
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
