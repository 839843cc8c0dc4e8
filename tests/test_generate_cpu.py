"""Generator mode (modarith_amd/generate.py), CPU side: the command-line rules of pseudo.py / monty.py for names and
refusals, and that a plug-in cross-compiles, loads and exports the whole per-prime C-ABI.  No compute (no GPU here).
The constants of the example moduli are checked against the reference's in tests/test_params.py, the generic oracle
against the reference-generated vectors of the same moduli in tests/test_generic_oracle.py."""
import ctypes
import os
import subprocess
import sys

import pytest

from modarith_amd import _lib, generate as gen


def test_tags_follow_the_generators_decoration_rule():
    # pseudo.py:1940-1944: an unnamed pseudo-Mersenne is tagged <n><m>; a named one keeps its name
    assert gen.resolve("2**255-19").name == "25519"
    assert gen.resolve("2**251-9").name == "2519" and gen.resolve("2**251-9").family == "pseudo"
    assert gen.resolve("X25519").name == "X25519"
    # monty.py:2510-2520: the PM shortcut makes the same tag, anything else must come with a name
    fp = gen.resolve("2**251-9", family="monty")
    assert fp.name == "2519" and fp.pm
    with pytest.raises(gen.GenerateError, match="must have a name"):
        gen.resolve("0xa9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377")
    fp = gen.resolve("BP256=0xa9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377")
    assert (fp.name, fp.family, fp.nlimbs, fp.radix) == ("BP256", "monty", 5, 52) and fp.ndash != 1
    # group orders: "00" + decimal (monty.py:2116-2118)
    q = 2**252 + 27742317777372353535851937790883648493
    assert gen.resolve("Q25519=00%d" % q).p == q


def test_refusals_match_the_reference():
    with pytest.raises(gen.GenerateError, match="64-bit"):
        gen.generate("2**255-19", wl=32)
    with pytest.raises(gen.GenerateError, match="sensible modulus"):          # pseudo.py:1563-1566: too small
        gen.resolve("2**89-1")
    with pytest.raises(gen.GenerateError, match="sensible modulus"):          # ... or not a prime
        gen.resolve("2**255-21")
    with pytest.raises(gen.GenerateError, match="exploitable pseudo-Mersenne"):   # pseudo.py:1590-1592
        gen.resolve("0xa9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377", family="pseudo")
    with pytest.raises(gen.GenerateError, match="starts with a digit"):       # pseudo.py:1554-1559
        gen.resolve("NOSUCHPRIME")
    with pytest.raises(gen.GenerateError, match="evaluated"):
        gen.resolve("2**255-19; import os")
    with pytest.raises(gen.GenerateError, match="C identifier"):
        gen.resolve("a-b=2**255-19")
    with pytest.raises(gen.GenerateError, match="built-in field"):            # a built-in name with another modulus
        gen.generate("X25519=2**251-9")
    # a built-in prime needs no plug-in
    g = gen.generate("X25519")
    assert g.tag == "X25519" and not g.built and g.lib == _lib.LIB_PATH


def test_plugin_cross_compiles_loads_and_exports_the_per_prime_abi(tmp_path):
    """2^130 - 5 (three 44-bit limbs) into a scratch directory: one hipcc unit; every <fn>_1305_batch and <fn>_1305_ct symbol
    that MODARITH_AMD_DECLARE(1305) declares is there, the main library is its only in-tree dependency, and a second call
    reuses it"""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmodarith_amd.so not built")
    g = gen.generate("2**130-5", plugin_dir=str(tmp_path))
    assert g.built and g.tag == "1305" and (g.params.nlimbs, g.params.radix) == (3, 44)
    lib = _lib.load_plugin("1305", g.lib)
    for fn in _lib.BATCH_FUNCS:
        assert hasattr(lib, "%s_1305_batch" % fn), fn
    for fn in _lib.SCALAR_FUNCS:
        assert hasattr(lib, "%s_1305_ct" % fn), fn
    needed = subprocess.run(["readelf", "-d", g.lib], capture_output=True, text=True).stdout
    assert "libmodarith_amd.so" in needed
    assert not gen.generate("2**130-5", plugin_dir=str(tmp_path)).built
    assert [m["tag"] for m in gen.installed(str(tmp_path))] == ["1305"]
    fp = gen.params_of_plugin("1305", str(tmp_path))
    assert (fp.p, fp.family) == (2**130 - 5, "pseudo")


def test_cli(tmp_path):
    env = dict(os.environ, MA_PLUGIN_DIR=str(tmp_path))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "modarith_amd.generate", "32", "2**255-19"], capture_output=True, text=True, cwd=root, env=env)
    assert p.returncode == 2 and "64-bit" in p.stdout
    p = subprocess.run([sys.executable, "-m", "modarith_amd.generate", "64"], capture_output=True, text=True, cwd=root, env=env)
    assert p.returncode == 2 and "Syntax error" in p.stdout
    p = subprocess.run([sys.executable, "-m", "modarith_amd.generate", "curve", "X", "edwards", "X25519", "2", "3", "5", "7", "11"], capture_output=True, text=True, cwd=root, env=env)
    assert p.returncode == 2 and "a = 1 and a = -1" in p.stdout
    p = subprocess.run([sys.executable, "-m", "modarith_amd.generate", "64", "X448"], capture_output=True, text=True, cwd=root, env=env)
    assert p.returncode == 0 and "Chosen radix is 56 bits, using 8 limbs" in p.stdout and "up to date" in p.stdout


def test_field_refuses_an_ungenerated_prime():
    torch = pytest.importorskip("torch")
    from modarith_amd.field import Field
    with pytest.raises(ValueError, match="neither built in"):
        Field("NOSUCH")


def test_generate_curve_refusals_and_cross_compile(tmp_path):
    """a curve that is not in curve.py's table (NIST P-224 over the built-in NIST224 field): the checks curve.py leaves to its
    user, then one hipcc unit whose plug-in exports the whole batched and scalar curve API"""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmodarith_amd.so not built")
    c = dict(next(x for x in gen.EXAMPLE_CURVES if x["name"] == "NIST224"))
    with pytest.raises(gen.GenerateError, match="not on the curve"):
        gen.generate_curve(**dict(c, gy=c["gy"] + 1), plugin_dir=str(tmp_path))
    with pytest.raises(gen.GenerateError, match="a = -3 and a = 0"):
        gen.generate_curve(**dict(c, a=2), plugin_dir=str(tmp_path))
    with pytest.raises(gen.GenerateError, match="built-in curve"):
        gen.generate_curve(**dict(c, name="NIST256"), plugin_dir=str(tmp_path))
    with pytest.raises(gen.GenerateError, match="neither built in nor generated"):
        gen.generate_curve(**dict(c, field="NOSUCH"), plugin_dir=str(tmp_path))
    with pytest.raises(gen.GenerateError, match="a = 1 and a = -1"):
        gen.generate_curve(**dict(c, kind="edwards", a=-3), plugin_dir=str(tmp_path))
    g = gen.generate_curve(**c, plugin_dir=str(tmp_path))
    assert g.built and (g.nlimbs, g.nbytes) == (4, 28)
    lib, nl, nb = _lib.load_curve_plugin("NIST224", g.lib)
    for fn in _lib.ED_BATCH_FUNCS:
        assert hasattr(lib, "ecn_nist224_%s_batch" % fn), fn
    for fn in _lib.ED_SCALAR_FUNCS:
        assert hasattr(lib, "ecn_nist224_%s" % fn), fn
    assert not gen.generate_curve(**c, plugin_dir=str(tmp_path)).built
    assert [m["curve"] for m in gen.installed_curves(str(tmp_path))] == ["NIST224"]


def test_generate_ladder_refusals_and_cross_compile(tmp_path):
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmodarith_amd.so not built")
    with pytest.raises(gen.GenerateError, match="TWIST_SECURE"):
        gen.generate_ladder("M383", "PM383", 516287, 3, twist_secure=False, plugin_dir=str(tmp_path))
    with pytest.raises(gen.GenerateError, match="COF is 2 or 3"):
        gen.generate_ladder("M383", "PM383", 516287, 4, plugin_dir=str(tmp_path))
    with pytest.raises(gen.GenerateError, match="64-bit words"):
        gen.generate_ladder("M224", "NIST224", 1234, 3, plugin_dir=str(tmp_path))        # 28-byte records
    with pytest.raises(gen.GenerateError, match="built in"):
        gen.generate_ladder("X25519", "X25519", 121665, 3, plugin_dir=str(tmp_path))
    lib = gen.generate_ladder("M383", "PM383", 516287, 3, plugin_dir=str(tmp_path))
    h = ctypes.CDLL(lib)
    assert hasattr(h, "rfc7748_M383") and hasattr(h, "rfc7748_M383_batch")
    assert gen.generate_ladder("M383", "PM383", 516287, 3, plugin_dir=str(tmp_path)) == lib
