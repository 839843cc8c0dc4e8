"""Sharding of independent batches across ranks.

Field elements / ladder records are independent units, so the path shards with no data-path
collective: rank g owns the contiguous slice [g*n/G, (g+1)*n/G).  The single collective on the path is
the final gather of result records (torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests).
Nothing here computes; it only partitions index ranges and moves finished results.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous, balanced slice [lo, hi) of n units for `rank` (first n % world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_records(local: torch.Tensor, n_total: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gather per-rank result records (rows of `local`, e.g. uint8 [n_local, 32]) onto rank `dst`
    in shard order; returns the [n_total, ...] tensor on dst and None elsewhere.  Ragged shards
    (n_total % world != 0) are padded to the largest shard for the collective and trimmed after."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d records, its shard has %d" % (rank, local.shape[0], sizes[rank]))
    m = max(sizes) if sizes else 0
    send = local
    if local.shape[0] != m:
        send = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
