"""Multi-GPU layout of the pricing path: shard the trade axis, reduce the aggregate once.

Trades are independent given the (tiny, replicated) curve tables, so each rank prices a contiguous
block of the trade axis chosen to balance *cash flows* (work is proportional to coupons, not trades;
SURVEY.md section 8(e)).  The only exchange is the aggregate ladder
``[pv, delta[P], gamma[P*P]]`` - one all-reduce of 1 + P + P*P doubles per curve over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  Per-trade results never move.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(flt_off: np.ndarray, fix_off: np.ndarray, world_size: int):
    """Contiguous trade ranges ``[(lo, hi)] * world_size`` with near-equal cash-flow counts."""
    n = int(flt_off.shape[0]) - 1
    work = (np.asarray(flt_off, dtype=np.int64) + np.asarray(fix_off, dtype=np.int64))   # cumulative flows
    total = int(work[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        cuts.append(int(np.searchsorted(work, target, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, n))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world_size)]


def shard_batch(batch, rank: int, world_size: int):
    """This rank's slice of a `TradeBatch`."""
    lo, hi = shard_bounds(batch.flt_off, batch.fix_off, world_size)[rank]
    return batch.slice(lo, hi), (lo, hi)


def allreduce_aggregate(agg, group=None):
    """In-place sum of the aggregate ladder tensor over the ranks (the single collective of the path)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(agg, op=dist.ReduceOp.SUM, group=group)
    return agg


def allgather_sum_fixed_order(agg, group=None):
    """Bit-stable alternative: gather every rank's partial and add them in rank order, so the result
    does not depend on the collective's internal reduction order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return agg
    parts = [torch.empty_like(agg) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, agg, group=group)
    total = parts[0].clone()
    for p in parts[1:]:
        total += p
    agg.copy_(total)
    return agg


# Aggregates that do not depend on the number of ranks.  A sum of per-rank ladders depends, in its last bits, on where the
# ranks' shards begin and end.  With CANONICAL chunks - the book cut once into `CANONICAL_CHUNKS` contiguous pieces of
# near-equal cash-flow count, whatever the world size (which must divide the chunk count: 1, 2, 3, 4, 6, 8, 12, 24) - every
# rank prices its run of chunks one by one (one aggregate ladder per chunk: a chunk's ladder is the same numbers on whichever
# rank prices it), the chunk ladders are all-gathered and every rank adds them in chunk order: one result, bit for bit, on
# 1, 2, 3 ... ranks.
CANONICAL_CHUNKS = 24


def canonical_chunks(flt_off: np.ndarray, fix_off: np.ndarray, rank: int, world_size: int, n_chunks: int = CANONICAL_CHUNKS):
    """``[(lo, hi)]``: the trade ranges of the canonical chunks this rank prices (chunks rank * n_chunks / world ...)."""
    if n_chunks % world_size:
        raise ValueError(f"world size {world_size} does not divide the {n_chunks} canonical chunks")
    per = n_chunks // world_size
    return shard_bounds(flt_off, fix_off, n_chunks)[rank * per:(rank + 1) * per]


def allgather_chunks_fixed_order(chunk_aggs, group=None):
    """``chunk_aggs [chunks of this rank, L]`` (tensor) -> the book ladder ``[L]``: all ranks' chunk ladders gathered (rank
    order = chunk order) and added one after the other, first chunk first - the same additions in the same order on every
    rank and for every world size."""
    import torch
    import torch.distributed as dist
    parts = [chunk_aggs]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [torch.empty_like(chunk_aggs) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, chunk_aggs.contiguous(), group=group)
    rows = torch.cat([p.reshape(-1, p.shape[-1]) for p in parts], dim=0)        # [all chunks, L], chunk order
    if rows.is_cuda:
        # one reduction kernel over a tensor whose shape ([CANONICAL_CHUNKS, L]) does not depend on the world size: the
        # same additions in the same order whatever the number of ranks
        return rows.sum(dim=0)
    total = torch.zeros_like(rows[0])
    for row in rows:
        total += row
    return total


def shard_by_work(work, world_size: int):
    """Contiguous ranges ``[(lo, hi)] * world_size`` of a list of items with per-item ``work`` (e.g. the coupons of
    each cross-currency swap), balanced like `shard_bounds`."""
    cum = np.concatenate(([0], np.cumsum(np.asarray(work, dtype=np.int64))))
    zeros = np.zeros_like(cum)
    return shard_bounds(cum, zeros, world_size)


def allreduce_book(aggregates: dict, group=None, device=None):
    """One all-reduce for a whole mixed book (BASELINE.json configs[4]): every ``agg_*`` entry of the result dicts
    of `price_batch` / `price_xccy_batch` (scalars and arrays; other keys are left alone) is packed into a single
    float64 buffer in sorted key order, summed over the ranks and unpacked in place.  ``device``: where the buffer
    lives - the rank's GPU for the RCCL backend, None (CPU) for gloo."""
    import torch
    keys = sorted(k for k in aggregates if k.startswith("agg_"))
    if not keys:
        return aggregates
    parts = [np.atleast_1d(np.asarray(aggregates[k], dtype=np.float64)).reshape(-1) for k in keys]
    buf = torch.from_numpy(np.concatenate(parts))
    if device is not None:
        buf = buf.to(device)
    allreduce_aggregate(buf, group)
    flat = buf.cpu().numpy()
    pos = 0
    for k, p in zip(keys, parts):
        v = flat[pos:pos + p.size].reshape(np.shape(aggregates[k]))
        aggregates[k] = float(v) if np.ndim(aggregates[k]) == 0 else v.copy()
        pos += p.size
    return aggregates
