"""Multi-rank path on CPU (gloo, world_size 2 and 3): shard arithmetic and the result gather of
modarith_amd/dist.py.  The per-rank compute stands in with the CPU oracle HERE ONLY (tests may use
the oracle; on GPUs each rank runs rfc7748_X25519_batch on its slice)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from modarith_amd.dist import gather_records, shard_range, shard_sizes


def test_shard_ranges_partition():
    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            lo_prev = 0
            for r in range(w):
                lo, hi = shard_range(n, r, w)
                assert lo == lo_prev and hi >= lo
                lo_prev = hi
            assert lo_prev == n
            s = shard_sizes(n, w)
            assert sum(s) == n and max(s) - min(s) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.oracle_binding import load_oracle
        from tests.util import vp
        oracle = load_oracle(build=False)
        rng = np.random.default_rng(42)                      # same global inputs on every rank
        k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        u = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        lo, hi = shard_range(n, rank, world)
        out = np.zeros((hi - lo, 32), dtype=np.uint8)
        if hi > lo:
            oracle.lib.batch_rfc7748_X25519(vp(np.ascontiguousarray(k[lo:hi])), vp(np.ascontiguousarray(u[lo:hi])), vp(out), hi - lo)
        got = gather_records(torch.from_numpy(out), n, dst=0)
        if rank == 0:
            want = np.zeros((n, 32), dtype=np.uint8)
            oracle.lib.batch_rfc7748_X25519(vp(k), vp(u), vp(want), n)
            q.put(bool(np.array_equal(got.numpy(), want)))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (3, 10), (2, 1)])
def test_gather_of_sharded_ladder_results(oracle, world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
