"""The paste-marker boundary as a shipped artefact (SURVEY 8(b) "What exists today": consumers take field.c by textual inclusion,
rfc7748.c:24-28, edwards.c:19-23, weierstrass.c:16-20, edge.c:5-9): include/field_<PRIME>.h, emitted by the parameter driver.  Without a
GPU: the headers are what the driver emits now, their macro block is the REFERENCE's macro block (the `#define` lines of the field.c
the reference generated, captured in tests/golden/field_<PRIME>.json params.header), all 32 names of pseudo.py:1413-1445 are mapped
to symbols the library exports, and a consumer written against the undecorated names alone compiles and links with gcc."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIMES = ("X25519", "NIST256", "X448")
NAMES = ("prop flatten modfsb modadd modsub modneg modmli modmul modsqr modcpy modnsqr modpro modinv nres redc modis1 modis0 modzer modone "
         "modint modqr modcmv modcsw modsqrt modshl modshr modhaf mod2r modexp modimp modsign modcmp").split()


@pytest.mark.parametrize("P", PRIMES)
def test_shim_header_is_current_and_carries_the_references_macro_block(P):
    from modarith_amd import emit
    from modarith_amd.params import derive
    path = os.path.join(ROOT, "include", "field_%s.h" % P)
    text = open(path).read()
    assert text == emit.field_shim_text(derive(P)), "include/field_%s.h is stale: python -m modarith_amd.emit" % P
    defs = [re.sub(r"\s+", " ", l).strip() for l in text.splitlines() if l.startswith("#define ") and "_H" not in l]
    block = [d for d in defs if not re.match(r"#define (\w+) \1_%s_ct$" % P, d)]
    ref = [re.sub(r"\s+", " ", l).strip() for l in load_golden("field_%s.json" % P)["params"]["header"]]
    assert sorted(block) == sorted(ref), (block, ref)                     # the same macros with the same values, nothing more
    mapped = [m.group(1) for d in defs for m in [re.match(r"#define (\w+) (\w+)_%s_ct$" % P, d)] if m and m.group(1) == m.group(2)]
    assert mapped == NAMES and len(NAMES) == 32                           # all 32, in the reference's emitted order
    assert emit.FIELD_C_NAMES == tuple(NAMES)


def test_every_mapped_name_is_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "modarith_amd", "libmodarith_amd.so"))
    for P in PRIMES:
        for fn in NAMES:
            assert hasattr(lib, "%s_%s_ct" % (fn, P)), "%s_%s_ct" % (fn, P)
            if fn not in ("modexp", "modimp"):
                assert hasattr(lib, "%s_%s_batch" % (fn, P))


@pytest.mark.parametrize("flag", ["", "-DUSE_X448"])
def test_consumer_uses_undecorated_names_only_and_links(flag, tmp_path):
    src = os.path.join(ROOT, "examples", "paste_marker_consumer.c")
    code = re.sub(r"/\*.*?\*/", "", open(src).read(), flags=re.S)
    assert not re.search(r"modarith_amd_|_ct\b|_batch\b", code), "the consumer must not know the library's names"
    for fn in "modimp modcpy modone modzer modcsw modadd modsub modsqr modmul modmli modpro modinv modexp".split():
        assert re.search(r"\b%s\(" % fn, code), fn
    assert "modmul(p2->z, E, p2->z)" in code and re.search(r"spint \w+\[Nlimbs\]", code)
    exe = str(tmp_path / "consumer")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", src, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "modarith_amd"), "-l:libmodarith_amd.so",
           "-Wl,-rpath," + os.path.join(ROOT, "modarith_amd"), "-o", exe] + ([flag] if flag else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    P = "X448" if flag else "X25519"
    und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    used = set(re.findall(r"\b(\w+)_%s_ct\b" % P, und))
    assert {"modimp", "modcsw", "modmul", "modsqr", "modmli", "modpro", "modinv", "modexp"} <= used and not re.search(r"rfc7748", und)


@pytest.mark.parametrize("P", PRIMES)
def test_all32_consumer_compiles_against_every_shim(P, tmp_path):
    """examples/paste_marker_all32.c (every one of the 32 names, undecorated) compiles and links against include/field_<P>.h for the three
    BASELINE fields; its run against the oracle-mapped twin is tests/test_gpu_paste_marker.py."""
    src = os.path.join(ROOT, "examples", "paste_marker_all32.c")
    code = re.sub(r"/\*.*?\*/", "", open(src).read(), flags=re.S)
    assert not re.search(r"modarith_amd_|_ct\b|_batch\b", code)
    for fn in NAMES:
        if fn == "modmli":
            continue                                                     # (guarded by MULBYINT, as in the reference's templates)
        assert re.search(r"\b%s\(" % fn, code), fn
    exe = str(tmp_path / "all32")
    r = subprocess.run(["gcc", "-O2", "-Wall", "-Werror", '-DFIELD_HEADER="field_%s.h"' % P, src, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "modarith_amd"),
                        "-l:libmodarith_amd.so", "-Wl,-rpath," + os.path.join(ROOT, "modarith_amd"), "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    used = set(re.findall(r"\b(\w+)_%s_ct\b" % P, und))
    assert len(used) >= 31, sorted(set(NAMES) - used)
