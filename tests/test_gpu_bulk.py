"""GPU vs the REFERENCE, wide: 3 input classes x 2^18 elements per prime through modmul / modsqr / modadd / modsub / modneg /
nres / redc / modmli on the HIP library, compared with the sha256 block digests of what the reference's own emitted C
returned for the same inputs (tests/golden/bulk_digests.json, made by tests/golden/make_bulk_digests.py in the build
container).  No oracle in between: inputs come from tests/util.py bulk_inputs, expected values from the reference."""
import numpy as np
import pytest

from tests.conftest import load_golden
from tests.util import BULK_CLASSES, BULK_OPS, block_digests, bulk_inputs, to_dev, to_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", ["X25519", "NIST256", "X448"])
def test_bulk_digests_gpu(P):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from modarith_amd.field import Field
    F = Field(P, tile=None)          # flat batches: the digests are taken over rows [N, n]
    g = load_golden("bulk_digests.json")
    n, blk = g["n"], g["block"]
    for cls in BULK_CLASSES:
        a, b = bulk_inputs(P, cls, n)
        A, B = to_dev(a), to_dev(b)
        if cls in ("uniform", "plus_p"):
            # the device-side generator of the same recipe yields the same limbs (bench.py's inputs)
            assert torch.equal(F.uniform(n, seed=42, array=100, plus_p=(cls == "plus_p")), A)
        for op in BULK_OPS:
            if op in ("modmul", "modadd", "modsub"):
                c = getattr(F, op)(A, B)
            elif op == "modmli_121665":
                c = F.modmli(A, 121665)
            else:
                c = getattr(F, op)(A)
            got = block_digests(to_np(c), blk)
            want = g["primes"][P][cls][op]
            bad = [k for k, (x, y) in enumerate(zip(got, want)) if x != y]
            assert not bad, "%s %s %s: %d of %d blocks differ from the reference (first: block %d)" % (P, cls, op, len(bad), len(want), bad[0])
        # the shared-multiplicand form against the reference's modmul: b0 = b[:, 0] broadcast (block 0 only carries a[j] * b[0]
        # for j = 0; compare through an explicit broadcast product instead)
        b0 = [int(v) for v in b[:, 0]]
        Bb = to_dev(np.ascontiguousarray(np.repeat(b[:, :1], n, axis=1)))
        assert torch.equal(F.modmuls(A, b0), F.modmul(A, Bb)), (P, cls, "modmuls")
