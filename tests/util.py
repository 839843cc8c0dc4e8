"""Shared helpers for the parity tests: seeded SoA batches and oracle batch calls (numpy in / out)."""
import ctypes

import numpy as np

from tests.oracle_binding import PRIMES


def random_soa(prime: str, n: int, seed: int, full_range: bool = True) -> np.ndarray:
    """uint64 [N, n]: every limb uniform in [0, 2^radix) and the top limb spanning all bits a value
    below 2^Nbits (< 2p) can set -- in-contract, non-canonical inputs (SURVEY 8(c) caveat 2)."""
    N, radix, nbits, _ = PRIMES[prime]
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << radix, size=(N, n), dtype=np.uint64)
    top_bits = nbits - radix * (N - 1)
    out[N - 1] = rng.integers(0, 1 << top_bits, size=n, dtype=np.uint64)
    return np.ascontiguousarray(out)


def vp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def oracle_bin(oracle, fn, prime, a, b):
    c = np.empty_like(a)
    oracle.fn("batch_" + fn, prime)(vp(a), vp(b), vp(c), a.shape[1], a.shape[1])
    return c


def oracle_un(oracle, fn, prime, a):
    c = np.empty_like(a)
    oracle.fn("batch_" + fn, prime)(vp(a), vp(c), a.shape[1], a.shape[1])
    return c


def oracle_mli(oracle, prime, a, k):
    c = np.empty_like(a)
    oracle.fn("batch_modmli", prime)(vp(a), int(k), vp(c), a.shape[1], a.shape[1])
    return c


def to_dev(a: np.ndarray):
    import torch
    return torch.from_numpy(a.view(np.int64)).cuda()


def to_np(t) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint64)


# ---- host model of the synthetic-input recipe (device side: csrc/kernels.h k_uniform)
_M64 = (1 << 64) - 1


def splitmix64_at(s0: int, t: int) -> int:
    z = (s0 + (t + 1) * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def uniform_model(p: int, nbits: int, seed: int, array: int, j: int) -> int:
    """the value of element j of stream (seed, array): ceil(nbits/64)+1 consecutive splitmix64 outputs, little-endian, mod p"""
    s0 = (seed * 0x9E3779B97F4A7C15 + array * 0xD1342543DE82EF95) & _M64
    nwd = (nbits + 63) // 64 + 1
    v = sum(splitmix64_at(s0, j * nwd + k) << (64 * k) for k in range(nwd))
    return v % p
