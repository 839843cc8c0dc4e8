#!/usr/bin/env python3
"""Wide comparison with the reference, committed as digests (SURVEY.md:359 "at least 10^4 seeded random pairs per prime").

Runs in the BUILD CONTAINER only (needs /root/reference): refgen.py drives the unmodified generators, gcc compiles the C
they emit, and a small batch harness (ours, below) loops the reference's own modmul / modsqr / modadd / modsub / modneg /
nres / redc / modmli over 2^18 elements per input class.  Only sha256 digests of the OUTPUT limbs are written
(tests/golden/bulk_digests.json, one 16-hex-digit digest per 4096-element block): data, no reference text.

Inputs are regenerated from tests/util.py bulk_inputs (a pure function of prime and class), so

  * tests/test_oracle_golden.py::test_bulk_digests_oracle   (CPU)  oracle outputs  -> same digests
  * tests/test_gpu_bulk.py::test_bulk_digests_gpu           (GPU)  HIP outputs     -> same digests, no oracle in between

  python tests/golden/make_bulk_digests.py
"""
import ctypes, json, os, sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refgen  # noqa: E402
import gio  # noqa: E402
from tests.util import BULK_BLOCK, BULK_CLASSES, BULK_N, BULK_OPS, block_digests, bulk_inputs  # noqa: E402

# harness (ours): element-major loops around the reference-emitted functions
BATCH = r"""
#include <stddef.h>
void bulk_bin(int op, const spint *a, const spint *b, spint *c, size_t n) {
    size_t j;
    for (j = 0; j < n; j++) {
        const spint *x = a + j * Nlimbs, *y = b + j * Nlimbs; spint *z = c + j * Nlimbs;
        if (op == 0) modmul(x, y, z); else if (op == 1) modadd(x, y, z); else modsub(x, y, z);
    }
}
void bulk_un(int op, const spint *a, spint *c, size_t n) {
    size_t j;
    for (j = 0; j < n; j++) {
        const spint *x = a + j * Nlimbs; spint *z = c + j * Nlimbs;
        if (op == 0) modsqr(x, z); else if (op == 1) modneg(x, z); else if (op == 2) nres(x, z);
        else if (op == 3) redc(x, z); else modmli(x, 121665, z);
    }
}
"""
BIN = {"modmul": 0, "modadd": 1, "modsub": 2}
UN = {"modsqr": 0, "modneg": 1, "nres": 2, "redc": 3, "modmli_121665": 4}


def main():
    out = {"n": BULK_N, "block": BULK_BLOCK, "digest": "sha256, first 16 hex digits, of the [Nlimbs, block] little-endian u64 slice of the output",
           "inputs": "tests/util.py bulk_inputs(prime, class)", "primes": {}}
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for script, P in (("pseudo.py", "X25519"), ("monty.py", "NIST256"), ("monty.py", "X448")):
        ns = refgen.load(script, 64, P)
        lib, _ = refgen.build(ns, BATCH, tag="bulk")
        lib.bulk_bin.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.bulk_un.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        per = {}
        for cls in BULK_CLASSES:
            a, b = bulk_inputs(P, cls)
            A, B = np.ascontiguousarray(a.T), np.ascontiguousarray(b.T)          # element-major for the reference's functions
            C = np.empty_like(A)
            d = {}
            for op in BULK_OPS:
                if op in BIN:
                    lib.bulk_bin(BIN[op], vp(A), vp(B), vp(C), A.shape[0])
                else:
                    lib.bulk_un(UN[op], vp(A), vp(C), A.shape[0])
                d[op] = block_digests(np.ascontiguousarray(C.T))
            per[cls] = d
            print(P, cls, "done", flush=True)
        out["primes"][P] = per
    gio.dump(out, "bulk_digests.json")
    print("wrote bulk_digests.json")


if __name__ == "__main__":
    main()
