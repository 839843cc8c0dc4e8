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


# ---- bulk reference comparison (tests/golden/bulk_digests.json; maker: tests/golden/make_bulk_digests.py)
# Inputs are a pure function of (prime, class): the maker (build container, reference-emitted C), the CPU test (oracle)
# and the GPU test (HIP library) all regenerate them from here and compare sha256 digests of the OUTPUT limbs.
BULK_N = 1 << 18
BULK_BLOCK = 4096
BULK_CLASSES = ("uniform", "plus_p", "edge")
BULK_OPS = ("modmul", "modsqr", "modadd", "modsub", "modneg", "nres", "redc", "modmli_121665")
_BULK_P = {"X25519": (1 << 255) - 19, "NIST256": (1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 1,
           "X448": (1 << 448) - (1 << 224) - 1}


def splitmix64_vec(s0: int, t: np.ndarray) -> np.ndarray:
    """splitmix64_at over a uint64 array of positions (wrapping 64-bit arithmetic)"""
    with np.errstate(over="ignore"):
        z = np.uint64(s0) + (t.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, array: int) -> int:
    return (seed * 0x9E3779B97F4A7C15 + array * 0xD1342543DE82EF95) & _M64


def uniform_soa(prime: str, n: int, seed: int, array: int, plus_p: bool = False) -> np.ndarray:
    """uint64 [N, n]: the moduniform recipe (uniform_model above, csrc/kernels.h k_uniform) for elements 0..n-1, as limbs
    with the top limb unmasked: canonical limbs of a value in [0,p), or of value + p when plus_p"""
    N, radix, nbits, _ = PRIMES[prime]
    p = _BULK_P[prime]
    nwd = (nbits + 63) // 64 + 1
    pos = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(nwd) + np.arange(nwd, dtype=np.uint64)[None, :])
    words = splitmix64_vec(_stream_key(seed, array), pos)                 # [n, nwd], little-endian words
    raw = words.astype("<u8").tobytes()
    out = np.empty((N, n), dtype=np.uint64)
    mask = (1 << radix) - 1
    for j in range(n):
        v = int.from_bytes(raw[j * nwd * 8:(j + 1) * nwd * 8], "little") % p
        if plus_p:
            v += p
        for i in range(N - 1):
            out[i, j] = v & mask
            v >>= radix
        out[N - 1, j] = v
    return out


def edge_soa(prime: str, n: int, seed: int, array: int) -> np.ndarray:
    """uint64 [N, n]: every limb drawn from the limb contract's edge classes {0, 1, 2^R-1, 2^R, 2^(R+1)-1, 2^(R+2)-1} (6 of 9
    draws) or uniform R-bit (3 of 9), by the splitmix64 stream (seed, array) at position limb * n + j"""
    N, R, _, _ = PRIMES[prime]
    edges = np.array([0, 1, (1 << R) - 1, 1 << R, (1 << (R + 1)) - 1, (1 << (R + 2)) - 1], dtype=np.uint64)
    pos = np.arange(N * n, dtype=np.uint64).reshape(N, n)
    r = splitmix64_vec(_stream_key(seed, array), pos)
    cls = (r % np.uint64(9)).astype(np.int64)
    rnd = (r >> np.uint64(8)) & np.uint64((1 << R) - 1)
    return np.ascontiguousarray(np.where(cls < 6, edges[np.minimum(cls, 5)], rnd))


def bulk_inputs(prime: str, cls: str, n: int = BULK_N):
    """(a, b) uint64 [N, n] for one input class of the bulk reference comparison"""
    if cls == "uniform":
        return uniform_soa(prime, n, 42, 100), uniform_soa(prime, n, 42, 101)
    if cls == "plus_p":
        return uniform_soa(prime, n, 42, 100, plus_p=True), uniform_soa(prime, n, 42, 101, plus_p=True)
    if cls == "edge":
        return edge_soa(prime, n, 43, 104), edge_soa(prime, n, 43, 105)
    raise ValueError(cls)


def block_digests(soa: np.ndarray, block: int = BULK_BLOCK):
    """sha256 (first 16 hex digits) of every `block`-element slice of a uint64 [N, n] batch, the slice taken limb-major
    ([N, block], little-endian words)"""
    import hashlib
    n = soa.shape[1]
    return [hashlib.sha256(np.ascontiguousarray(soa[:, k:k + block]).astype("<u8").tobytes()).hexdigest()[:16] for k in range(0, n, block)]


# ------------------------------------------------------------------ generator mode (modarith_amd/generate.py)
def generated_tags():
    """tags of the unnamed example moduli the generator mode builds plug-ins for (fixtures: tests/golden/field_<TAG>.json)"""
    from modarith_amd import generate as gen
    return [gen.resolve(arg, fam).name for arg, fam in gen.EXAMPLES]


def derive_any(P: str):
    """FieldParams of a built-in prime or of a generated example modulus, by tag"""
    from modarith_amd import generate as gen
    from modarith_amd.params import derive
    for arg, fam in gen.EXAMPLES:
        fp = gen.resolve(arg, fam)
        if fp.name == P:
            return fp
    return derive(P)
