#!/usr/bin/env python3
"""Where can the window table of the bit-exact k_ed_mul / k_ed_mul2 live?  (round 6, review item: "take the window tables out of global memory,
or show it cannot pay")

  python tools/table_placement.py build      (CPU: builds the variants of ED25519 through tools/ecn_exp.hip -> tools/ecn_exp_tp*_ED25519.bin)
  python tools/table_placement.py run        (GPU box: runs them, prints rates, limb digests and the shader clock under each kernel)

The table of ecnXXXmul (edwards.c:435-449) is nine entries x three coordinates x five 64-bit limbs = 1 080 bytes per lane, 69 KB per
wave, and the constant-time select (edwards.c:382-401) reads every entry at every one of the 64 windows: 61 KB per scalar.  Its limbs
are the reference's (a pseudo-Mersenne product's output limbs depend on its input limbs, so entries cannot be stored canonical or
packed), and the window width is the reference's.  Variants, all at the limbs of the tree (digests must agree) except `none`:

  glob   the tree: per-wave slabs in the global workspace, [entry][coord][limb][64 lanes], four waves per SIMD
  lds    the slab in LDS: 69 KB per wave -> TWO waves per CU fit the 160 KB (against sixteen)
  none   no table at all: get() hands back a register value, put() stores nothing -- WRONG results, the arithmetic alone: what any
         placement whatsoever could reach at most
  wps2   the tree at two waves per SIMD (table workspace 141 MB: inside the 256 MiB Infinity Cache), for the clock

Registers cannot hold it: eight non-trivial entries are 240 32-bit words per lane beside the ~117 the arithmetic needs (a 512-register
wave -- one per SIMD -- measured 20 % slower in round 3 before a table went anywhere)."""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "modarith_amd", "csrc")
TOOLS = os.path.join(ROOT, "tools")
MUL_CLASS = "-DCURVE_MUL_CLASS=ma::Edwards<ma::C_ED25519, ma::FieldH51<ma::P_X25519>>"


def patched(tag, edits):
    d = os.path.join("/tmp", "ma_tp_" + tag)
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(CSRC, d)
    p = os.path.join(d, "curve.h")
    s = open(p).read()
    for a, b in edits:
        assert s.count(a) >= 1, (tag, a)
        s = s.replace(a, b)
    open(p, "w").write(s)
    return d


def build():
    none = patched("none", [
        ("F::unpack(q[(0 * N + I) * 64], w.x, I);", "F::unpack((((spint)(k + 1) * 0x9E3779B1ull + (spint)I) & 0x3ffffffffffffull) + (q == nullptr), w.x, I);"),
        ("F::unpack(q[(1 * N + I) * 64], w.y, I);", "F::unpack((((spint)(k + 2) * 0x85EBCA6Bull + (spint)I) & 0x3ffffffffffffull), w.y, I);"),
        ("F::unpack(q[(2 * N + I) * 64], w.z, I);", "F::unpack((((spint)(k + 3) * 0xC2B2AE35ull + (spint)I) & 0x3ffffffffffffull), w.z, I);"),
        ("q[(0 * N + I) * 64] = F::pack(w.x, I);", "(void)q;"),
        ("q[(1 * N + I) * 64] = F::pack(w.y, I);", ""),
        ("q[(2 * N + I) * 64] = F::pack(w.z, I);", ""),
    ])
    lds = patched("lds", [
        ("    __shared__ signed char digs[E::NDIG * 64];\n    const typename E::Table W{ws + (size_t)blockIdx.x * E::SLAB_WORDS, threadIdx.x};",
         "    __shared__ signed char digs[E::NDIG * 64];\n    __shared__ spint lds_slab[9 * E::ENTRY_WORDS];\n    const typename E::Table W{lds_slab, threadIdx.x};"),
    ])
    for tag, src, wps in (("tpglob", CSRC, 4), ("tpwps2", CSRC, 2), ("tpnone", none, 4), ("tplds", lds, 1)):
        cmd = ["bash", os.path.join(TOOLS, "build_ecn_exp.sh"), tag, src, "ED25519", "-DUSE_FH51", MUL_CLASS, "-DMA_MUL_WPS=%d" % wps]
        print(" ".join(cmd[:5]), "...", flush=True)
        subprocess.run(cmd, check=True)
        assert os.path.exists(os.path.join(TOOLS, "ecn_exp_%s_ED25519.bin" % tag))


def run():
    for tag in ("tpglob", "tpwps2", "tplds", "tpnone", "tpglob"):
        b = os.path.join(TOOLS, "ecn_exp_%s_ED25519.bin" % tag)
        ops = "1" if tag == "tplds" else "3"                                  # (the LDS slab holds the one table of mul)
        p = subprocess.run([b, "20", ops], capture_output=True, text=True, timeout=600)
        sys.stdout.write(p.stdout + p.stderr[-500:])
        sys.stdout.flush()


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "build"]()
