"""Fused chains (modarith_amd/fuse.py), CPU side: what is emitted, that it cross-compiles for gfx950 and exports its entry
point.  No compute (no GPU here)."""
import ctypes
import os

import pytest

from modarith_amd import _lib
from modarith_amd.fuse import Chain


def _accept(P, name="accept"):
    ch = Chain(P, name)
    x, y = ch.inputs(2)
    s = ch.modsqr(ch.modmul(ch.modadd(x, y), ch.modsub(x, y)))
    ch.output(ch.modinv(s))
    return ch


def test_source_is_the_call_sequence_on_registers():
    ch = _accept("X25519")
    src = ch.source()
    body = src[src.index("void body("):src.index("template <int EPT>")]
    calls = [l.strip() for l in body.splitlines() if l.strip().startswith("F::")]
    assert calls == ["F::modadd(v0, v1, v2);", "F::modsub(v0, v1, v3);", "F::modmul(v2, v3, v4);", "F::modsqr(v4, v5);",
                     "F::modinv(v5, nullptr, v6); inv_normalise<F>(v6);"]
    assert "HEAVY = true" in src and "load_soa<P, EPT>(A.in[1], L, t, v1);" in src and "store_soa<P, EPT>(A.out[0], L, t, v6);" in src
    assert ch.traffic_bytes() == 120 and ch.unfused_traffic_bytes() == 520          # two arrays in, one out; five round trips
    assert ch.symbol == "chain_accept_X25519_batch"


def test_builder_refusals():
    with pytest.raises(ValueError, match="C identifier"):
        Chain("X25519", "9lives")
    with pytest.raises(ValueError, match="neither built in nor generated"):
        Chain("NOSUCH", "c")
    a, b = Chain("X25519", "a"), Chain("X25519", "b")
    x = a.input()
    with pytest.raises(ValueError, match="values of this chain"):
        b.modsqr(x)
    a.modsqr(x)
    with pytest.raises(ValueError, match="before the first operation"):
        a.input()
    with pytest.raises(ValueError, match="at least one input and one output"):
        a.source()
    with pytest.raises(ValueError, match="C int"):
        a.modmli(x, 1 << 40)


def test_chain_cross_compiles_and_exports(tmp_path):
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmodarith_amd.so not built")
    ch = Chain("NIST256", "twoout")
    x, y = ch.inputs(2)
    t, w = ch.modadd(x, y), ch.modsub(x, y)
    ch.output(ch.modmul(t, w))
    ch.output(ch.modmli(ch.modsqr(t), 121665))
    f = ch.build(plugin_dir=str(tmp_path))
    assert f.built and os.path.exists(f.path) and hasattr(f.lib, "chain_twoout_NIST256_batch")
    assert "HEAVY = false" in ch.source()
    assert not ch.build(plugin_dir=str(tmp_path)).built          # cached by content


def test_text_front_end():
    """python -m modarith_amd.fuse <prime> <name> "<statements>": the same chain as the builder calls"""
    from modarith_amd.fuse import parse
    a = _accept("X25519").source()
    b = parse("X25519", "accept", "in x, y; t = modadd(x, y); w = modsub(x, y); s = modsqr(modmul(t, w)); out modinv(s)").source()
    assert a == b
    c = parse("X448", "sw", "in g, f; sel d; g2, f2 = modcsw(d, g, f); out modmli(g2, 39081), f2")
    assert (c.nin, c.nsel, len(c.outs)) == (2, 1, 2) and "F::modcsw(s0, v2, v3);" in c.source() and "F::modmli(v2, 39081, v4);" in c.source()
    for bad, msg in (("in x; out nosuch(x)", "unknown operation"), ("in x; out y", "unknown value"), ("in x; x + 1", "expected"),
                     ("in x; a, b = modsqr(x); out a", "does not produce")):
        with pytest.raises(ValueError, match=msg):
            parse("X25519", "bad", bad)
