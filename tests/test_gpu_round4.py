"""Round-4 additions on the GPU box: the single-process multi-device C example, the host-resident pipelined
ladder (SURVEY 8(d) "H2D/D2H for C5"), which ladder kernel an entry point launches (MA_LADDER_IMPL), the default layout of
Field, the strong-scaling switch of bench.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_multi_gpu_shard_example_bytes_equal_oracle(tmp_path):
    """examples/multi_gpu_shard.c: one process, one host thread per device (modarith_amd_device_count() of them: one on this
    box, eight on the node the driver scales on), contiguous shards of one host batch, results in one host buffer -- every
    byte against the CPU oracle's rfc7748()"""
    from tests.oracle_binding import load_oracle
    from tests.util import vp
    exe = str(tmp_path / "multi_gpu_shard")
    libdir = os.path.join(ROOT, "modarith_amd")
    subprocess.check_call(["gcc", "-O2", "-pthread", os.path.join(ROOT, "examples", "multi_gpu_shard.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", libdir, "-l:libmodarith_amd.so", "-Wl,-rpath," + libdir, "-o", exe])
    out = str(tmp_path / "records.bin")
    lg = 14
    p = subprocess.run([exe, str(lg), "0", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    from modarith_amd import _lib
    ndev = _lib.load().modarith_amd_device_count()
    assert ("devices %d records %d" % (ndev, 1 << lg)) in p.stdout, p.stdout
    assert p.stdout.count("RFC 7748 vector ok") == ndev
    raw = np.fromfile(out, dtype=np.uint8)
    n = 1 << lg
    assert raw.size == 3 * n * 32
    bk, bu, bv = (np.ascontiguousarray(raw[i * n * 32:(i + 1) * n * 32].reshape(n, 32)) for i in range(3))
    want = np.empty_like(bv)
    oracle = load_oracle()
    oracle.lib.oracle_parallel(3, vp(bk), vp(bu), vp(want), n, 0, max(1, len(os.sched_getaffinity(0))))
    assert np.array_equal(bv, want)


def test_multi_gpu_shard_example_eight_threads_on_one_device(tmp_path):
    """round 5: the same program with eight shards / eight host threads whatever the number of GPUs (--oversubscribe: thread i on
    device i mod count).  On this one-GPU box all eight drive the library's per-device staging buffer (eight scalar rfc7748() calls
    at once), its scratch pool and their own streams concurrently; the bytes must be those of the one-thread run and of the oracle."""
    from tests.oracle_binding import load_oracle
    from tests.util import vp
    exe = str(tmp_path / "multi_gpu_shard")
    libdir = os.path.join(ROOT, "modarith_amd")
    subprocess.check_call(["gcc", "-O2", "-pthread", os.path.join(ROOT, "examples", "multi_gpu_shard.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", libdir, "-l:libmodarith_amd.so", "-Wl,-rpath," + libdir, "-o", exe])
    lg, n = 15, 1 << 15
    out8, out1 = str(tmp_path / "r8.bin"), str(tmp_path / "r1.bin")
    p8 = subprocess.run([exe, str(lg), "8", out8, "--oversubscribe"], capture_output=True, text=True, timeout=300)
    assert p8.returncode == 0, p8.stdout + p8.stderr
    assert p8.stdout.count("RFC 7748 vector ok") == 8 and "oversubscribed: 8 shards" in p8.stdout, p8.stdout
    bounds = sorted((int(a), int(b)) for a, b in __import__("re").findall(r"records \[(\d+), (\d+)\) ok", p8.stdout))
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[i][1] == bounds[i + 1][0] for i in range(7))      # the shards partition the batch
    p1 = subprocess.run([exe, str(lg), "1", out1], capture_output=True, text=True, timeout=300)
    assert p1.returncode == 0, p1.stdout + p1.stderr
    r8, r1 = np.fromfile(out8, dtype=np.uint8), np.fromfile(out1, dtype=np.uint8)
    assert r8.size == 3 * n * 32 and np.array_equal(r8, r1)
    bk, bu, bv = (np.ascontiguousarray(r8[i * n * 32:(i + 1) * n * 32].reshape(n, 32)) for i in range(3))
    want = np.empty_like(bv)
    load_oracle().lib.oracle_parallel(3, vp(bk), vp(bu), vp(want), n, 0, max(1, len(os.sched_getaffinity(0))))
    assert np.array_equal(bv, want)


def test_host_resident_pipelined_ladder_equals_device_resident():
    from modarith_amd.field import rfc7748
    from modarith_amd.hostio import PinnedBytes, ladder_host
    for curve, nb in (("X25519", 32), ("X448", 56)):
        n = 3 * 8192 + 77                                    # three full chunks and a ragged one
        rng = np.random.default_rng(11)
        hk, hu, hv = PinnedBytes(n, nb), PinnedBytes(n, nb), PinnedBytes(n, nb)
        hk.array[:] = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
        hu.array[:] = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
        hu.array[5] = 0                                      # a low-order input among them
        ladder_host(curve, hk, hu, hv, chunk=8192)
        want = rfc7748(curve, torch.from_numpy(hk.array.copy()).cuda(), torch.from_numpy(hu.array.copy()).cuda()).cpu().numpy()
        assert np.array_equal(hv.array, want), curve
        for h in (hk, hu, hv):
            h.close()


def _child(code, env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return p.stdout


_LADDER_PROBE = r'''
import numpy as np, torch
from modarith_amd import _lib
from modarith_amd.field import rfc7748
lib = _lib.load()
rng = np.random.default_rng(5)
n = 8192 + 64
k = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
u = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
o = rfc7748("X25519", k, u)
print("PY", lib.modarith_amd_last_launch().decode())
ws = torch.empty(lib.rfc7748_X25519_batch_workspace_bytes(n) // 8 + 1, dtype=torch.int64, device="cuda")
o2 = torch.empty_like(o)
_lib.check(lib.rfc7748_X25519_batch_ws(k.data_ptr(), u.data_ptr(), o2.data_ptr(), n, ws.data_ptr(), ws.numel() * 8, None), "ws")
print("WS", lib.modarith_amd_last_launch().decode())
torch.cuda.synchronize()
print("EQ", bool(torch.equal(o, o2)))
import hashlib
print("H", hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())
'''


def test_ladder_impl_switch_is_honoured_by_every_entry_point():
    """MA_LADDER_IMPL=field (README: A/B and parity runs of the field.c-form ladder) must select that kernel for the Python
    wrapper at n >= 8192 and for rfc7748_<C>_batch_ws as well -- not only for the plain C entry"""
    a = _child(_LADDER_PROBE, {})
    b = _child(_LADDER_PROBE, {"MA_LADDER_IMPL": "field"})
    la = dict(l.split(" ", 1) for l in a.strip().splitlines())
    lb = dict(l.split(" ", 1) for l in b.strip().splitlines())
    assert la["PY"] == "rfc7748(split)" and la["WS"] == "rfc7748(split)", la
    assert lb["PY"] == "rfc7748(field form)" and lb["WS"] == "rfc7748(field form)", lb
    assert la["EQ"] == "True" and lb["EQ"] == "True" and la["H"] == lb["H"]       # same bytes whichever kernel ran


def test_field_default_layout_is_tiled_and_partial_tiles_are_flat():
    from modarith_amd.field import Field
    F = Field("X25519")
    assert F.tile == 4096
    a = F.uniform(8192, seed=1, array=0)
    assert a.dim() == 3 and tuple(a.shape) == (2, 5, 4096)
    # every allocating method agrees with from_limbs for sizes that are not whole tiles: flat, nothing raises
    for n in (4096, 4097, 8192 + 5, 3 * 4096 + 1):
        x = F.uniform(n, seed=1, array=0)
        assert x.dim() == 2 and x.shape[1] == n
        y = F.uniform(n, seed=1, array=1)
        assert F.modmul(x, y).shape == x.shape
        assert F.modzer(n).shape == x.shape and F.modone(n).shape == x.shape
        assert F.from_limbs(F.to_limbs(x)).shape == x.shape
    Ff = Field("X25519", tile=None)
    b = Ff.uniform(8192, seed=1, array=0)
    assert b.dim() == 2 and torch.equal(Ff.to_flat(a), b)
    assert torch.equal(F.to_flat(F.modsqr(a)), Ff.modsqr(b))


def test_bench_strong_scaling_switch_and_new_keys(tmp_path):
    """--scaling strong divides 2^MA_BENCH_LOG2_LADDER_TOTAL records over the ranks (BASELINE configs[4] literally: 2^26 over 8);
    here 2^20 over the one rank of this box.  Also the round-4 keys of the record (since round 6 in the detail file; the line the
    driver parses is the compact one, tests/test_gpu_bench.py)."""
    detail = str(tmp_path / "detail.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MA_BENCH_LOG2_LADDER_TOTAL="20", MA_BENCH_LOG2_ELEMS="22", MA_BENCH_LOG2_X448="16", MA_BENCH_DETAIL=detail)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu", "--no-others", "--no-traffic", "--scaling", "strong"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["x25519"]["scaling"] == "strong" and line["x25519"]["host_resident_pipelined_per_s"] > 0
    d = json.load(open(detail))
    assert d["scaling"] == "weak"                                  # the headline modmul line
    x = d["x25519"]
    assert x["scaling"] == "strong" and x["scalars_total"] == 1 << 20 and x["scalars_per_gpu"] == 1 << 20
    h = x["host_resident"]
    assert h["h2d_ms"] > 0 and h["d2h_ms"] > 0 and h["end_to_end_pipelined_per_s"] > 0.25 * h["end_to_end_serial_per_s"]      # (a sanity bound, not a rate: host threads of a shared box)
    assert d["x448"]["value"] > 1e6 and d["x448"]["roofline"]["bound"] == "valu-mad"
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    assert d["launch_stats"]["launches"] >= 20


@pytest.mark.parametrize("C,name,radix,n", [("ed25519", "ED25519", 51, 1200), ("ed448", "ED448", 56, 300)])
def test_scalar_multiplication_on_non_canonical_representatives(oracle, C, name, radix, n):
    """the API's element contract (SURVEY 8c caveat 2): any representative below 2p, limbs possibly over Radix bits.  Points
    whose coordinates are such representatives -- value + p where that stays below 2p, and carries pushed DOWN so that lower
    limbs exceed their radix by up to two bits while the integer stays the same -- through ecn mul / mul2(exact): projective
    limbs equal to the oracle's (for ED25519 this goes through the half-limb resident field, csrc/fh51.h, with excess bits in
    every upper half)."""
    import ctypes
    from modarith_amd.edwards import Edwards
    from modarith_amd.params import derive
    Ed = Edwards(name)
    nb, nl = Ed.nbytes, Ed.N
    fp = derive("X25519" if name == "ED25519" else "X448")
    rng = np.random.default_rng(29)
    k0 = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    e = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    f = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    base = Ed.mul(torch.from_numpy(k0).cuda(), Ed.gen(n)).cpu().numpy().view(np.uint64).copy()       # [3, nl, n], limbs of the reference
    mask = (1 << radix) - 1
    for c in range(3):
        for j in range(n):
            limbs = [int(v) for v in base[c, :, j]]
            if name == "ED25519":                             # plain field: the other representative of the same residue
                v = sum(l << (radix * i) for i, l in enumerate(limbs))
                if j % 3 == 0 and v + fp.p < 2 * fp.p:
                    v += fp.p
                    limbs = [(v >> (radix * i)) & mask for i in range(nl - 1)] + [v >> (radix * (nl - 1))]
            if j % 2 == 0:                                    # same integer, lower limbs over Radix bits (borrow 1..3 from the limb above)
                for i in range(nl - 1):
                    take = min(int(rng.integers(1, 4)), limbs[i + 1])
                    limbs[i + 1] -= take
                    limbs[i] += take << radix
            base[c, :, j] = np.array(limbs, dtype=np.uint64)
    assert int(base.max()) < 1 << (radix + 2)
    P = torch.from_numpy(base.view(np.int64)).cuda()
    want = base.reshape(3 * nl, n).copy()
    oracle.ecn(C, "batch_mul")(e.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), n, n)
    got = Ed.mul(torch.from_numpy(e).cuda(), P.clone()).cpu().numpy().view(np.uint64).reshape(3 * nl, n)
    assert np.array_equal(got, want)
    # ... and through the fused kernels, which re-pack the limbs into 32-bit words (from51 / from56 + a weak carry): the bytes of
    # get(mul(e, P)) -- the limbs of that product were just compared with the oracle's
    et = torch.from_numpy(e).cuda()
    ft = torch.from_numpy(f).cuda()
    xw, yw, _ = Ed.get(Ed.mul(et, P.clone()))
    xg, yg, _ = Ed.mul_get(et, P)
    assert torch.equal(xg, xw) and torch.equal(yg, yw)
    D = Ed.dbl(P.clone())
    x2, y2, _ = Ed.get(Ed.mul2(et, P, ft, D))
    xg, yg, _ = Ed.mul2_get(et, P, ft, D)
    assert torch.equal(xg, x2) and torch.equal(yg, y2)
    x3, y3, _ = Ed.get(Ed.mul2(et, Ed.gen(n), ft, P))
    xg, yg, _ = Ed.mulgen2_get(et, ft, P)
    assert torch.equal(xg, x3) and torch.equal(yg, y3)
    # the same points through add and dbl (element-wise kernels, limb form) and through the exact double multiplication
    Q = Ed.dbl(P.clone())
    R = Ed.mul2(torch.from_numpy(e).cuda(), P, torch.from_numpy(f).cuda(), Q, exact=True).cpu().numpy().view(np.uint64)
    Pt, _ = oracle.ed[C]
    for j in range(0, n, max(1, n // 24)):
        p, q, r = Pt(), Pt(), Pt()
        for c, nm in enumerate(("x", "y", "z")):
            for i in range(nl):
                getattr(p, nm)[i] = int(base[c, i, j])
        oracle.ecn(C, "cpy")(ctypes.byref(p), ctypes.byref(q))
        oracle.ecn(C, "dbl")(ctypes.byref(q))
        oracle.ecn(C, "mul2")(e[j].ctypes.data_as(ctypes.c_char_p), ctypes.byref(p), f[j].ctypes.data_as(ctypes.c_char_p), ctypes.byref(q), ctypes.byref(r))
        for c, nm in enumerate(("x", "y", "z")):
            assert [int(v) for v in R[c, :, j]] == list(getattr(r, nm)), (j, nm)
