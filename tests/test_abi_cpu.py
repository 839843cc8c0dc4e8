"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports every symbol
include/modarith_amd.h declares; host-side argument checking; no compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from modarith_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "modarith_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(modarith_amd_\w+)\s*\(", text))
    names.discard("modarith_amd_")
    macro = text[text.index("#define MODARITH_AMD_DECLARE(P)"):text.index("MODARITH_AMD_DECLARE(X25519)")]
    per_prime = re.findall(r"\b(\w+)_##P##_(ct|batch)\s*\(", macro)
    primes = re.findall(r"^MODARITH_AMD_DECLARE\((\w+)\)", text, flags=re.M)
    for fn, kind in per_prime:
        for P in primes:
            names.add("%s_%s_%s" % (fn, P, kind))
    names.update(re.findall(r"\b(rfc7748_\w+)\s*\(", text))
    emacro = text[text.index("#define MODARITH_AMD_DECLARE_EDWARDS(c, NL)"):text.index("MODARITH_AMD_DECLARE_EDWARDS(ed25519, 5)")]
    for fn in re.findall(r"\becn_##c##_(\w+)\s*\(", emacro):
        for c in re.findall(r"^MODARITH_AMD_DECLARE_EDWARDS\((\w+),", text, flags=re.M):
            names.add("ecn_%s_%s" % (c, fn))
    names.update(re.findall(r"\b(ecn_\w+_mul(?:2|gen|gen2)?_get_(?:batch|workspace_bytes))\s*\(", text))
    return sorted(names), primes


def test_every_declared_symbol_is_exported(lib):
    names, primes = _declared_symbols()
    from modarith_amd import emit
    assert primes == list(emit.BUILT_PRIMES)
    assert len(names) > 900
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_tables_cover_header(lib):
    from modarith_amd import _lib
    names, _ = _declared_symbols()
    bound = {"%s_%s_batch" % (f, P) for f in _lib.BATCH_FUNCS for P in _lib.PRIMES}
    bound |= {"%s_%s_ct" % (f, P) for f in _lib.SCALAR_FUNCS for P in _lib.PRIMES}
    bound |= set(_lib.UTIL_FUNCS) | {"rfc7748_X25519", "rfc7748_X448", "rfc7748_X25519_batch", "rfc7748_X448_batch",
                                     "rfc7748_X25519_batch_ws", "rfc7748_X448_batch_ws", "rfc7748_X25519_batch_workspace_bytes", "rfc7748_X448_batch_workspace_bytes"}
    bound |= {"ecn_%s_%s_batch" % (c, f) for c in _lib.CURVES for f in _lib.ED_BATCH_FUNCS}
    bound |= {"ecn_%s_%s" % (c, f) for c in _lib.CURVES for f in _lib.ED_SCALAR_FUNCS}
    bound |= set(_lib.FUSED_FUNCS)
    assert bound == set(names)


def test_field_info_matches_driver(lib):
    from modarith_amd.params import derive
    from modarith_amd import emit
    for P in emit.BUILT_PRIMES:
        vals = [ctypes.c_int() for _ in range(5)]
        assert lib.modarith_amd_field_info(P.encode(), *[ctypes.byref(v) for v in vals]) == 1
        fp = derive(P)
        assert [v.value for v in vals] == [fp.nlimbs, fp.radix, fp.n, fp.nbytes, int(fp.montgomery)]
    assert lib.modarith_amd_field_info(b"NOPE", None, None, None, None, None) == 0
    assert lib.modarith_amd_abi_version() == 2


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from modarith_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    """the product package must not reference the oracle (SURVEY/task rule: oracle is test infrastructure)"""
    pkg = os.path.join(ROOT, "modarith_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                text = open(os.path.join(root, f)).read()
                assert "liboracle" not in text and "oracle_binding" not in text and "from tests" not in text, f


def test_device_is_normalised_to_an_indexed_form():
    """Field(p, "cuda") / Curve(name, torch.device("cuda")) must bind to the CURRENT device with its index:
    torch.device("cuda") != torch.device("cuda:0"), so the index-less form stored verbatim rejected every tensor."""
    import torch
    from modarith_amd.field import normalise_device
    cur = lambda: 3
    for d in (None, "cuda", torch.device("cuda")):
        assert normalise_device(d, current=cur) == torch.device("cuda:3")
    assert normalise_device("cuda:1", current=cur) == torch.device("cuda:1")
    assert normalise_device(torch.device("cuda", 5), current=cur) == torch.device("cuda:5")
    assert normalise_device(2, current=cur) == torch.device("cuda:2")
    with pytest.raises(ValueError):
        normalise_device("cpu", current=cur)


def test_tiled_layout_helpers_follow_the_documented_address_map():
    """tile_batch / flatten_batch (torch ops, any device) against the address formula of include/modarith_amd.h:
    limb i of element j at buf[((j // tile) * N + i) * tile + j % tile]"""
    import torch
    from modarith_amd.field import flatten_batch, tile_batch
    N, tile, n = 5, 128, 3 * 128
    flat = torch.arange(N * n, dtype=torch.int64).reshape(N, n)            # value = i * n + j
    t = tile_batch(flat, tile)
    assert t.shape == (3, N, tile) and t.is_contiguous()
    buf = t.reshape(-1)
    for i in (0, 2, 4):
        for j in (0, 1, 127, 128, 200, n - 1):
            assert int(buf[((j // tile) * N + i) * tile + j % tile]) == i * n + j
    assert torch.equal(flatten_batch(t), flat) and flatten_batch(flat) is flat
    with pytest.raises(ValueError):
        tile_batch(flat, 100)
    with pytest.raises(ValueError):
        tile_batch(flat[:, :200], 128)


def test_recommended_layout_helpers(lib):
    """modarith_amd_recommended_ld / modarith_amd_batch_words (pure host functions): tiles of 4096 from two whole tiles on,
    flat rows below; the word count of a batch in either form -- and Field's own default follows the same recommendation"""
    rec, words = lib.modarith_amd_recommended_ld, lib.modarith_amd_batch_words
    assert rec(0) == 0 and rec(1) == 1 and rec(4096) == 4096 and rec(8191) == 8191
    assert rec(8192) == 4096 and rec(8193) == 4096 and rec(1 << 24) == 4096
    assert words(100, 5, 100) == 500 and words(100, 5, 128) == 640            # flat: nlimbs * ld
    assert words(8192, 5, 4096) == 2 * 5 * 4096                               # two whole tiles
    assert words(8193, 5, 4096) == 3 * 5 * 4096                               # a partial last tile is a whole tile of storage
    assert words(1 << 24, 8, 4096) == 8 << 24
    assert words(10, 0, 10) == 0 and words(10, 5, 0) == 0
    from modarith_amd.field import Field
    assert Field.DEFAULT_TILE == rec(1 << 24)
    F = Field.__new__(Field)                                                  # layout rule only: no device needed
    F.tile = Field.DEFAULT_TILE
    assert [F.creates_tiled(n) for n in (1, 4096, 8191, 8192, 8193, 12288, 1 << 24)] == [False, False, False, True, False, True, True]
    F.tile = None
    assert not F.creates_tiled(1 << 24)


def test_last_launch_and_scratch_trim_are_callable_without_a_gpu(lib):
    assert lib.modarith_amd_last_launch() in (b"",) or isinstance(lib.modarith_amd_last_launch(), bytes)


def test_scalar_entry_points_record_device_errors_instead_of_aborting():
    """Up to round 4 a scalar _ct call whose staging buffer could not be made (no device, no memory) called abort() inside the shared
    library.  Now it records the failure -- modarith_amd_last_error() for the thread, the sticky modarith_amd_status() for the
    process, modarith_amd_thread_status() for the calling thread --, launches nothing, hands back zero-filled PURE outputs, leaves
    operands that are also inputs as they were (round 6: modnsqr(a, n), modmul(z, e, z)), and its predicates answer -1.  Here: no GPU at all (a child process, so that a regression
    that aborts fails this test instead of ending the test run)."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
from modarith_amd import _lib
L = _lib.load()
L.modarith_amd_status.restype = ctypes.c_int
L.modarith_amd_last_error.restype = ctypes.c_char_p
assert L.modarith_amd_status() == 0
U = ctypes.c_uint64 * 5
a, b, c = U(1, 2, 3, 4, 5), U(5, 4, 3, 2, 1), U(9, 9, 9, 9, 9)
L.modmul_X25519_ct(a, b, c)                      # void modmul(const spint*, const spint*, spint*): nothing to return an error through
st = L.modarith_amd_status()
msg = L.modarith_amd_last_error().decode()
L.modis0_X25519_ct.restype = ctypes.c_int
r = L.modis0_X25519_ct(a)
bk = (ctypes.c_char * 32)(); bv = (ctypes.c_char * 32)(*([b"x"] * 32))
L.rfc7748_X25519(bk, bk, bv)
z = U(7, 7, 7, 7, 7)
L.modmul_X25519_ct(z, b, z)                      # output aliases an input: untouched
L.modnsqr_X25519_ct(a, 3)                        # in place: untouched
L.modarith_amd_thread_status.restype = ctypes.c_int
ts = L.modarith_amd_thread_status()
print("status", st, "c", list(c), "pred", r, "bv", bytes(bv) == bytes(32), "z", list(z), "a", list(a), "thread", ts == st, "msg", msg)
L.modarith_amd_clear_status()
L.modarith_amd_clear_thread_status()
assert L.modarith_amd_status() == 0 and L.modarith_amd_thread_status() == 0
''' % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    out = p.stdout.strip().splitlines()[-1]
    assert out.startswith("status ") and int(out.split()[1]) != 0, out
    assert "c [0, 0, 0, 0, 0]" in out and "pred -1" in out and "bv True" in out and "staging" in out, out
    assert "z [7, 7, 7, 7, 7]" in out and "a [1, 2, 3, 4, 5]" in out and "thread True" in out, out


def test_fused_workspace_sizes_are_bounded(lib):
    """round 5: the ladder / Straus forms of the fused Edwards kernels work in chunks of 2^20 records, so what a caller must allocate stops
    growing there (include/modarith_amd.h): 140 / 236 bytes per record for ed25519 / ed448, plus the table slabs of the resident grid
    (2048 waves) for the double multiplications"""
    sz = ctypes.c_size_t
    for name in ("ecn_ed25519_mul_get_workspace_bytes", "ecn_ed25519_mulgen2_get_workspace_bytes", "ecn_ed25519_mul2_get_workspace_bytes",
                 "ecn_ed448_mul_get_workspace_bytes", "ecn_ed448_mulgen2_get_workspace_bytes", "ecn_ed448_mul2_get_workspace_bytes"):
        f = getattr(lib, name)
        f.argtypes, f.restype = [sz], sz
    chunk = 1 << 20
    for n in (1, 4096, chunk - 1, chunk, chunk + 1, 1 << 26):
        m = min(n, chunk)
        assert lib.ecn_ed25519_mul_get_workspace_bytes(n) == 140 * m == lib.ecn_ed25519_mulgen2_get_workspace_bytes(n)
        assert lib.ecn_ed448_mul_get_workspace_bytes(n) == 236 * m == lib.ecn_ed448_mulgen2_get_workspace_bytes(n)
        waves = min((m + 63) // 64, 2048)
        # (+ 127: the table slab is aligned to 128 bytes INSIDE the workspace, whatever address the caller passes -- round 6)
        assert lib.ecn_ed25519_mul2_get_workspace_bytes(n) == waves * 64 * 18 * 128 + 140 * m + 127
        assert lib.ecn_ed448_mul2_get_workspace_bytes(n) == waves * 64 * 18 * 256 + 236 * m + 127
