"""GPU parity, round 2: what GPUTEST_r01 did not execute.

  * modnsqr at the boundary against the reference's own outputs (tests/golden/field_<P>_r2.json) and the oracle;
  * OUT-OF-CONTRACT limbs (>= 2^(Radix+2), up to all-ones) against the reference's outputs: the exact 128-bit product
    path, which the default policy must fall back to (csrc/kernels.h OpMulAuto & co.);
  * mixed waves: batches in which most lanes are in contract (-> split products) and some lanes / whole waves are not
    (-> the wave votes for the exact products), against the oracle, through modmul / modsqr / nres / redc;
  * the three product policies (default per-wave vote, MA_FORCE_EXACT=1, MA_FORCE_FAST=1 -- process-static switches):
    the parity files re-run in child processes under each forced policy;
  * BASELINE configs 2-4 at their FULL 2^24 size against the oracle, every element (not a sample).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import limbs, load_golden
from tests.oracle_binding import PRIMES
from tests.util import oracle_bin, oracle_un, random_soa, to_dev, to_np, vp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORE = ["X25519", "NIST256", "X448"]
EXTRA = list(__import__("modarith_amd.emit", fromlist=["EXTRA_PRIMES"]).EXTRA_PRIMES) + __import__("tests.util", fromlist=["generated_tags"]).generated_tags()
FORCED_FAST = os.environ.get("MA_FORCE_FAST") == "1"      # unguarded split products: only defined inside the limb contract


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module", params=CORE + EXTRA)
def ctx2(request, torch_cuda):
    from modarith_amd.field import Field
    P = request.param
    return P, Field(P), load_golden("field_%s_r2.json" % P)


def test_golden_modnsqr(ctx2):
    """modnsqr(a, k): k squarings in place (pseudo.py:745-755, monty.py:1182-1192), vs the reference's outputs"""
    P, F, g = ctx2
    recs = g["modnsqr"]
    for k in sorted({r["k"] for r in recs}):
        sel = [r for r in recs if r["k"] == k]
        a = F.from_limbs([limbs(r["a"]) for r in sel])
        ret = F.modnsqr(a, k)
        assert ret.data_ptr() == a.data_ptr()                       # in place, as in the reference
        assert F.to_limbs(a) == [limbs(r["out"]) for r in sel], (P, k)


@pytest.mark.parametrize("P", CORE)
def test_modnsqr_batch_vs_oracle(oracle, torch_cuda, P):
    n = 4099
    a = random_soa(P, n, 31)
    for k in (0, 1, 4, 17):
        want = a.copy()
        f = oracle.fn("modnsqr", P)
        N = a.shape[0]
        for j in range(0, n, 7):                                     # the scalar oracle, a strided sample of lanes
            z = oracle.arr(P, [int(v) for v in a[:, j]])
            f(z, k)
            want[:, j] = list(z)
        from modarith_amd.field import Field
        got = to_np(Field(P).modnsqr(to_dev(a), k))
        assert np.array_equal(got[:, ::7], want[:, ::7]), (P, k)
        # and against k single squarings on the device for every lane
        F = Field(P)
        x = to_dev(a)
        for _ in range(k):
            x = F.modsqr(x)
        assert np.array_equal(got, to_np(x)), (P, k)


@pytest.mark.skipif(FORCED_FAST, reason="MA_FORCE_FAST=1 runs the split products unguarded; they are only defined inside the limb contract")
def test_golden_out_of_contract_limbs(ctx2):
    """limbs far beyond the excess budget: field.c has no error path, its 64-bit wrap-around defines the answer, and the
    engine must return the same words (default policy: the wave votes for the exact products)"""
    P, F, g = ctx2
    recs = g["ooc"]
    A = F.from_limbs([limbs(r["a"]) for r in recs])
    B = F.from_limbs([limbs(r["b"]) for r in recs])
    for op in ("modmul", "modadd", "modsub"):
        assert F.to_limbs(getattr(F, op)(A, B)) == [limbs(r[op]) for r in recs], (P, op)
    for op in ("modsqr", "nres", "redc", "modneg"):
        assert F.to_limbs(getattr(F, op)(A)) == [limbs(r[op]) for r in recs], (P, op)
    assert F.to_limbs(F.modmli(A, 121665)) == [limbs(r["modmli_121665"]) for r in recs], (P, "modmli")
    # shared-multiplicand product with an out-of-contract common operand
    b0 = limbs(recs[5]["b"])
    want = F.to_limbs(F.modmul(A, F.from_limbs([b0] * len(recs))))
    assert F.to_limbs(F.modmuls(A, b0)) == want


def mixed_batch(P, n, seed):
    """mostly in-contract lanes; every 97th lane, one whole 256-element block and the last 3 lanes carry limbs from
    the out-of-contract classes (2^(R+2), 2^(R+3)-1, 2^63, 2^64-1, random 64-bit)"""
    N, R, _, _ = PRIMES[P]
    rng = np.random.default_rng(seed)
    a = random_soa(P, n, seed)
    classes = np.array([1 << (R + 2), (1 << (R + 3)) - 1, 1 << 63, (1 << 64) - 1, (1 << (R + 2)) - 1, 0], dtype=np.uint64)
    bad = np.zeros(n, dtype=bool)
    bad[::97] = True
    bad[4096:4352] = True
    bad[-3:] = True
    idx = np.nonzero(bad)[0]
    pick = classes[rng.integers(0, len(classes), size=(N, idx.size))]
    rnd = rng.integers(0, 1 << 64, size=(N, idx.size), dtype=np.uint64)
    a[:, idx] = np.where(rng.integers(0, 3, size=(N, idx.size)) == 0, rnd, pick)
    return a, bad


@pytest.mark.skipif(FORCED_FAST, reason="out-of-contract lanes are outside the domain of the unguarded split products")
@pytest.mark.parametrize("P", CORE)
def test_mixed_waves_vs_oracle(oracle, torch_cuda, P):
    """the per-wave policy vote of OpMulAuto / OpSqrAuto / OpNresAuto / OpRedcAuto: waves whose lanes are all in
    contract take the split products, waves holding even one out-of-contract limb take the exact ones; every lane must
    equal the oracle either way.  n is odd (scalar tail launch) and > 2 * 64 * 256 so that both kinds of wave occur."""
    from modarith_amd.field import Field
    F = Field(P)
    n = (1 << 16) + 3
    a, bad_a = mixed_batch(P, n, 41)
    b, bad_b = mixed_batch(P, n, 42)
    b = np.ascontiguousarray(np.roll(b, 13, axis=1))
    assert bad_a.sum() > 900 and (~bad_a).reshape(-1)[:8192].sum() > 7000
    A, B = to_dev(a), to_dev(b)
    assert np.array_equal(to_np(F.modmul(A, B)), oracle_bin(oracle, "modmul", P, a, b)), "modmul"
    assert np.array_equal(to_np(F.modsqr(A)), oracle_un(oracle, "modsqr", P, a)), "modsqr"
    assert np.array_equal(to_np(F.nres(A)), oracle_un(oracle, "nres", P, a)), "nres"
    assert np.array_equal(to_np(F.redc(A)), oracle_un(oracle, "redc", P, a)), "redc"
    # shared multiplicand (k_mul_shared: the vote is on a[] alone, b0 is checked on the host): an in-contract b0 from a clean
    # lane (split products in the clean waves, exact ones in the others) and an out-of-contract b0 (exact everywhere)
    for j in (int(np.nonzero(~bad_b)[0][7]), int(np.nonzero(bad_b)[0][3])):
        b0 = [int(v) for v in np.roll(b, -13, axis=1)[:, j]]
        bb = np.ascontiguousarray(np.repeat(np.array(b0, dtype=np.uint64)[:, None], n, axis=1))
        assert np.array_equal(to_np(F.modmuls(A, b0)), oracle_bin(oracle, "modmul", P, a, bb)), "modmuls (b0 from lane %d)" % j
        got = torch_cuda.empty_like(A)
        F.modmuls(A[:, 1:], b0, out=got[:, 1:])
        assert np.array_equal(to_np(got)[:, 1:], oracle_bin(oracle, "modmul", P, a, bb)[:, 1:]), "modmuls (unaligned)"
    # unaligned view (8-byte-per-lane kernels) takes the same vote
    out = torch_cuda.empty_like(A)
    F.modmul(A[:, 1:], B[:, 1:], out=out[:, 1:])
    assert np.array_equal(to_np(out)[:, 1:], oracle_bin(oracle, "modmul", P, a, b)[:, 1:]), "modmul (unaligned)"


@pytest.mark.parametrize("P", CORE)
def test_contract_edge_classes_vs_oracle(oracle, torch_cuda, P):
    """every limb from the contract's edge classes {0, 1, 2^R-1, 2^R, 2^(R+1)-1, 2^(R+2)-1, random}: all lanes in
    contract, so the default policy and MA_FORCE_FAST=1 run the split / chain products at their proven bounds
    (emit.split_point, emit.chain_ok); MA_FORCE_EXACT=1 runs the 128-bit ones.  All must equal the oracle."""
    from modarith_amd.field import Field
    F = Field(P)
    N, R, _, _ = PRIMES[P]
    n = 1 << 16
    rng = np.random.default_rng(43)
    edges = np.array([0, 1, (1 << R) - 1, 1 << R, (1 << (R + 1)) - 1, (1 << (R + 2)) - 1], dtype=np.uint64)

    def draw():
        cls = rng.integers(0, 9, size=(N, n))
        rnd = rng.integers(0, 1 << R, size=(N, n), dtype=np.uint64)
        return np.ascontiguousarray(np.where(cls < 6, edges[np.minimum(cls, 5)], rnd))
    a, b = draw(), draw()
    a[:, :6] = edges[None, :]                                          # the same class in every limb
    b[:, :6] = edges[None, ::-1]
    A, B = to_dev(a), to_dev(b)
    assert np.array_equal(to_np(F.modmul(A, B)), oracle_bin(oracle, "modmul", P, a, b)), "modmul"
    assert np.array_equal(to_np(F.modsqr(A)), oracle_un(oracle, "modsqr", P, a)), "modsqr"
    assert np.array_equal(to_np(F.nres(A)), oracle_un(oracle, "nres", P, a)), "nres"
    assert np.array_equal(to_np(F.redc(A)), oracle_un(oracle, "redc", P, a)), "redc"
    for j in range(6):                                                 # shared multiplicand: each edge class as the common operand
        b0 = [int(v) for v in b[:, j]]
        bb = np.ascontiguousarray(np.repeat(b[:, j:j + 1], n, axis=1))
        assert np.array_equal(to_np(F.modmuls(A, b0)), oracle_bin(oracle, "modmul", P, a, bb)), "modmuls class %d" % j


@pytest.mark.parametrize("P", CORE + ["NIST521", "GM384", "NIST224", "ED25519Q"])
def test_uniform_inputs_match_host_model(torch_cuda, P):
    """the benchmark's input recipe is regenerable on the host from (seed, array, j) alone"""
    from modarith_amd.field import Field
    from tests.util import derive_any as derive
    from tests.util import uniform_model
    F, fp = Field(P), derive(P)
    n = 1031
    for seed, array, first in ((42, 0, 0), (42, 1, 0), (7, 3, (1 << 33) + 5)):
        got = F.to_ints(F.uniform(n, seed=seed, array=array, first=first))
        lim = F.to_limbs(F.uniform(n, seed=seed, array=array, first=first))
        want = [uniform_model(fp.p, fp.n, seed, array, first + j) for j in range(n)]
        assert got == want, (P, seed, array, first)
        assert lim == [fp.to_limbs(v) for v in want]                       # canonical limbs, top limb included
        shifted = F.to_limbs(F.uniform(n, seed=seed, array=array, first=first, plus_p=True))
        assert shifted == [fp.to_limbs(v + fp.p) for v in want]            # [p,2p), top limb unmasked
    # a window of a longer stream equals the same positions generated alone
    a = F.uniform(4096, array=2)
    b = F.uniform(100, array=2, first=1000)
    assert torch_cuda.equal(a[:, 1000:1100], b)
    assert len(set(F.to_ints(a))) > 4000


# ---------------------------------------------------------------- BASELINE configs 2-4, full size, every element
FULL = [("X25519", "modmul", 0), ("NIST256", "modmul", 1), ("X448", "modmul", 2), ("X448", "modsqr", 7)]


@pytest.mark.skipif(os.environ.get("MA_POLICY_CHILD") == "1", reason="full-size runs belong to the parent suite")
@pytest.mark.parametrize("P,op,kind", FULL)
def test_full_size_vs_oracle(oracle, torch_cuda, P, op, kind):
    """2^24 elements (BASELINE.json configs[1..3]), all of them against the CPU oracle on all host cores; operands
    are in Montgomery / internal form (nres of uniform values below 2^Nbits, itself checked on the way)"""
    torch = torch_cuda
    from modarith_amd.field import Field
    F = Field(P)
    N, R, nbits, _ = PRIMES[P]
    n = 1 << 24
    cores = len(os.sched_getaffinity(0))
    g = torch.Generator(device="cuda").manual_seed(2400 + kind)

    def raw():
        t = torch.randint(0, 1 << R, (N, n), dtype=torch.int64, device="cuda", generator=g)
        t[N - 1] &= (1 << (nbits - R * (N - 1))) - 1
        return t
    A = F.nres(raw())
    ha = np.ascontiguousarray(A.cpu().numpy().view(np.uint64))
    want = np.empty_like(ha)
    if op == "modmul":
        B = F.nres(raw())
        hb = np.ascontiguousarray(B.cpu().numpy().view(np.uint64))
        got = F.modmul(A, B)
        assert oracle.lib.oracle_parallel(kind, vp(ha), vp(hb), vp(want), n, n, cores) == 0
        del B, hb
    else:
        got = F.modsqr(A)
        assert oracle.lib.oracle_parallel(kind, vp(ha), vp(ha), vp(want), n, n, cores) == 0
    hg = got.cpu().numpy().view(np.uint64)
    assert np.array_equal(hg, want), "%s %s differs from the oracle at full size" % (P, op)
    # fed back once more (outputs as inputs: what time.c and the ladder do), on the first 2^22 lanes
    m = 1 << 22
    g2 = F.modmul(got[:, :m], A[:, :m])                     # views of the wide batches: the result takes their limb stride
    w2 = np.empty((N, m), dtype=np.uint64)
    hgm, ham = np.ascontiguousarray(hg[:, :m]), np.ascontiguousarray(ha[:, :m])
    mulkind = {"X25519": 0, "NIST256": 1, "X448": 2}[P]
    assert oracle.lib.oracle_parallel(mulkind, vp(hgm), vp(ham), vp(w2), m, m, cores) == 0
    assert np.array_equal(g2.cpu().numpy().view(np.uint64), w2)


# ---------------------------------------------------------------- the forced product policies, in child processes
@pytest.mark.skipif(os.environ.get("MA_POLICY_CHILD") == "1", reason="already inside a forced-policy child")
@pytest.mark.parametrize("knob", ["MA_FORCE_EXACT", "MA_FORCE_FAST"])
def test_parity_suite_under_forced_policy(knob):
    """MA_FORCE_EXACT / MA_FORCE_FAST are read once per process (csrc/capi_common.hip), so each forced policy gets its
    own pytest child over the field parity files; the child must load the HIP library and pass everything it runs"""
    env = dict(os.environ, MA_POLICY_CHILD="1")
    env.pop("MA_FORCE_EXACT", None)
    env.pop("MA_FORCE_FAST", None)
    env[knob] = "1"
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_round2.py", "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = p.stdout[-1500:] + p.stderr[-500:]
    assert p.returncode == 0, tail
    last = [l for l in p.stdout.splitlines() if " passed" in l][-1]
    assert int(last.split(" passed")[0].split()[-1]) > 300, tail


# ---------------------------------------------------------------- curve soak as a test (tools/soak.py curves, 2^13 scalars)
ALL_CURVES = ["ED25519", "NIST256", "ED448", "NIST384", "SECP256K1", "NUMS256W", "NUMS256E", "ED248", "ED376", "NIST521", "ED500"]


@pytest.mark.skipif(os.environ.get("MA_POLICY_CHILD") == "1", reason="curve kernels do not read the product-policy switches")
@pytest.mark.parametrize("name", ALL_CURVES)
def test_curve_soak_projective_limbs(oracle, torch_cuda, name):
    """2^13 random scalars x random projective points per curve: the fused bit-exact `ecn mul` against the restated
    edwards.c / weierstrass.c on the CPU, projective limbs compared (for ED25519 this runs the half-limb products)"""
    torch = torch_cuda
    from modarith_amd.edwards import Curve
    n = 1 << 13
    C = Curve(name)
    g = torch.Generator(device="cuda").manual_seed(77)
    e0 = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    P = C.mul(e0, C.gen(n))
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    e[:16] = 0
    e[16:32] = 255
    hp = np.ascontiguousarray(P.cpu().numpy().view(np.uint64)).reshape(3 * C.N, n)
    he = np.ascontiguousarray(e.cpu().numpy())
    want = hp.copy()
    getattr(oracle.lib, "ecn_%s_batch_mul" % name.lower())(vp(he), vp(want), n, n)
    got = C.mul(e, P.clone()).cpu().numpy().view(np.uint64).reshape(3 * C.N, n)
    assert np.array_equal(got, want), name


@pytest.mark.skipif(os.environ.get("MA_POLICY_CHILD") == "1", reason="no product policy involved")
@pytest.mark.parametrize("name", ["ED25519", "ED448", "NIST256", "NIST521", "SECP256K1"])
def test_limb_budget_predicate(torch_cuda, name):
    """modlimbs / Curve.limbs_ok: every output of the library keeps the limb budget (< 2^(Radix+2)) the curve kernels
    rely on, and a fabricated limb at or above it is flagged, lane by lane"""
    torch = torch_cuda
    from modarith_amd.edwards import Curve
    from modarith_amd.field import Field
    from modarith_amd import curves
    n = 1000
    C = Curve(name)
    up = name.upper()
    F = Field((curves.CURVES[up] if up in curves.CURVES else curves.W_CURVES[up]).field)
    g = torch.Generator(device="cuda").manual_seed(5)
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    P = C.mul(e, C.gen(n))
    assert bool(C.limbs_ok(P).all())
    assert bool(C.limbs_ok(C.add(P.clone(), C.dbl(P.clone()))).all())
    bad = P.clone()
    lanes = [0, 63, 64, 500, n - 1]
    for k, j in enumerate(lanes):
        bad[k % 3, (k * 2) % C.N, j] = (1 << (F.radix + 2)) + k          # one limb exactly at / just above the bound
    bad[1, 0, 7] = -1                                                    # all-ones limb
    ok = C.limbs_ok(bad).cpu().numpy()
    want = np.ones(n, dtype=np.int32)
    want[lanes + [7]] = 0
    assert np.array_equal(ok, want)
    a = F.uniform(n, seed=3)
    a[F.N - 1, 5] = (1 << (F.radix + 2)) - 1                             # the largest limb inside the budget
    assert bool(F.modlimbs(a).all())


# ---------------------------------------------------------------- coarse rate floors (catch codegen regressions, not tuning)
RATE_FLOORS = {"ED25519": 2.0e7, "NIST256": 9e6, "ED448": 5e6, "NIST384": 3e6, "SECP256K1": 1.2e7, "NUMS256W": 1.0e7, "NUMS256E": 1.8e7,
               "ED248": 1.7e7, "ED376": 7e6, "NIST521": 1.2e6, "ED500": 2.2e6}


@pytest.mark.skipif(os.environ.get("MA_POLICY_CHILD") == "1", reason="timing belongs to the parent suite")
@pytest.mark.parametrize("name", ALL_CURVES)
def test_curve_mul_rate_floor(torch_cuda, name):
    """`ecn mul` on one resident grid (2^17 points) must reach 40 % of the documented rate: a register-allocation accident in
    one of these 250-VGPR kernels (spills inside the window loop) costs a factor, not per cents -- NIST P-521 ran 8x
    slower for a round without any test noticing"""
    import time
    torch = torch_cuda
    from modarith_amd.edwards import Curve
    C = Curve(name)
    n = 1 << 17
    g = torch.Generator(device="cuda").manual_seed(5)
    e = torch.randint(0, 256, (n, C.nbytes), dtype=torch.uint8, device="cuda", generator=g)
    P = C.gen(n)
    C.mul(e[:4096].contiguous(), C.gen(4096))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        Q = P.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        C.mul(e, Q)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert n / best >= RATE_FLOORS[name], "%s ecn mul %.3g/s, floor %.3g/s" % (name, n / best, RATE_FLOORS[name])


def test_index_less_device_forms(torch_cuda):
    """Field(p, "cuda") and Curve(name, torch.device("cuda")) bind to the current device and accept their own tensors"""
    torch = torch_cuda
    from modarith_amd.edwards import Curve
    from modarith_amd.field import Field
    for d in ("cuda", torch.device("cuda"), None, 0):
        F = Field("X25519", d)
        assert F.device == torch.device("cuda", torch.cuda.current_device())
        a = F.uniform(100, seed=1)
        assert torch.equal(F.modmul(a, a), F.modsqr(a))
        C = Curve("ED25519", d)
        G = C.gen(8)
        assert bool(C.limbs_ok(C.dbl(G)).all())


@pytest.mark.skipif(os.environ.get("MA_POLICY_CHILD") == "1", reason="no product policy involved")
@pytest.mark.parametrize("P", CORE + EXTRA)
def test_modlimbs_predicate_every_prime(torch_cuda, P):
    """modlimbs against a host model for every built prime: 1 iff every limb < 2^(Radix+2); vacuously 1 for radix >= 62
    (GM384: 2^64 is not a 64-bit value -- a shift by 64 used to make the kernel answer 0 for every non-zero element);
    for the 2^256-189 field (NUMS256W, whose curve kernels run the folded half-limb products) the bound is (2^64-1)/mm."""
    from modarith_amd.field import Field
    F = Field(P)
    R, N = F.radix, F.N
    n = 4099
    rng = np.random.default_rng(77)
    top = min(R + 2, 64)
    edges = [0, 1, (1 << R) - 1, 1 << R, (1 << top) - 1, (1 << 64) - 1, 1 << 63]
    if top < 64:
        edges += [1 << top, (1 << top) + 1]
    limit = (1 << top) - 1
    if P == "NUMS256W":
        limit = ((1 << 64) - 1) // 0xbd0
        edges += [limit, limit + 1]
    e = np.array(edges, dtype=np.uint64)
    a = rng.integers(0, 1 << R, size=(N, n), dtype=np.uint64)
    pick = rng.integers(0, 3 * len(e), size=(N, n))
    a = np.where(pick < len(e), e[np.minimum(pick, len(e) - 1)], a)
    a[:, 0] = 0
    a[:, 1] = np.uint64(limit)
    want = (a <= np.uint64(limit)).all(axis=0).astype(np.int32)
    assert 0 < want.sum() <= n
    got = F.modlimbs(to_dev(np.ascontiguousarray(a))).cpu().numpy()
    assert np.array_equal(got, want), P
