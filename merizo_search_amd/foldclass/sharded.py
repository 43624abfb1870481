"""Row-sharded database search across the GPUs of one node.

The reference has no multi-GPU search (faiss.index_cpu_to_all_gpus replicates each block,
reference dbsearch.py:228-230).  Here rank s of S holds the contiguous rows
[s*ceil(N/S), (s+1)*ceil(N/S)) resident in its HBM -- global row = shard offset + local row,
exactly the reference's `I += i0` block logic (dbsearch.py:238-242) -- every rank scans its
shard for the same (replicated) query batch, and ONE collective exchanges the per-shard
results: an all-gather of 12*nq*k bytes per rank (float32 score + int64 row), over RCCL/xGMI
when the process group's backend is "nccl".  Every rank then merges the S sorted lists
(ResultHeap.add_result / finalize, dbsearch.py:240-245), so all ranks hold identical results.

One process per GPU (torch.distributed); no collective touches the database itself.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) held by `rank`: contiguous blocks of ceil(n_total/world) rows."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def pack_results(scores, idx):
    """(float32 [nq,k], int64 [nq,k]) -> one uint8 buffer of 12*nq*k bytes (single collective)."""
    import torch

    return torch.cat([scores.contiguous().view(torch.uint8).reshape(-1), idx.contiguous().view(torch.uint8).reshape(-1)])


def unpack_results(buf, world: int, nq: int, k: int):
    """uint8 [world, 12*nq*k] -> (float32 [world,nq,k], int64 [world,nq,k])."""
    import torch

    ns = 4 * nq * k
    buf = buf.reshape(world, -1)
    scores = buf[:, :ns].contiguous().view(torch.float32).reshape(world, nq, k)
    idx = buf[:, ns:].contiguous().view(torch.int64).reshape(world, nq, k)
    return scores, idx


def allgather_results(scores, idx, group=None):
    """All-gather every rank's [nq,k] results -> ([S,nq,k], [S,nq,k]) on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return scores.unsqueeze(0), idx.unsqueeze(0)
    world = dist.get_world_size(group)
    nq, k = scores.shape
    mine = pack_results(scores, idx)
    out = torch.empty((world, mine.numel()), dtype=torch.uint8, device=mine.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, mine, group=group)          # one RCCL all-gather over xGMI
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out = torch.stack(parts)
    return unpack_results(out, world, nq, k)


class PackedExchange:
    """Preallocated buffers for the per-batch exchange of a row-sharded search (serving loop).

    ``out_s`` / ``out_i`` are views into this rank's packed block -- the local top-k is written
    straight into it (ops.ip_topk_finish / ops.ip_topk outputs) -- ``exchange()`` is ONE
    all-gather of that block, and ``merge()`` reads the S blocks in place
    (ms_topk_merge_strided).  Per batch: no allocation, no pack / unpack copies.
    """

    def __init__(self, nq: int, k: int, device, group=None):
        import torch
        import torch.distributed as dist

        self.nq, self.k, self.group = nq, k, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.idx_offset = (4 * nq * k + 7) // 8 * 8                 # int64 rows start 8-byte aligned
        self.block_bytes = self.idx_offset + 8 * nq * k
        self.mine = torch.zeros(self.block_bytes, dtype=torch.uint8, device=device)
        self.out_s = self.mine[: 4 * nq * k].view(torch.float32).reshape(nq, k)
        self.out_i = self.mine[self.idx_offset:].view(torch.int64).reshape(nq, k)
        self.gathered = torch.empty((self.world, self.block_bytes), dtype=torch.uint8, device=device)
        self.merged_s = torch.empty((nq, k), dtype=torch.float32, device=device)
        self.merged_i = torch.empty((nq, k), dtype=torch.int64, device=device)

    def exchange(self):
        import torch
        import torch.distributed as dist

        if self.world == 1:
            self.gathered[0].copy_(self.mine)
        elif dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(self.gathered, self.mine, group=self.group)      # one RCCL all-gather over xGMI
        else:
            parts = [torch.empty_like(self.mine) for _ in range(self.world)]
            dist.all_gather(parts, self.mine, group=self.group)
            self.gathered.copy_(torch.stack(parts))
        return self.gathered

    def blocks(self):
        """Views ([S,nq,k] float32, [S,nq,k] int64) of the gathered blocks (strided, no copy)."""
        import torch

        nqk = self.nq * self.k
        s = self.gathered[:, : 4 * nqk].view(torch.float32).reshape(self.world, self.nq, self.k)
        i = self.gathered[:, self.idx_offset:].view(torch.int64).reshape(self.world, self.nq, self.k)
        return s, i

    def merge(self, merge_fn: Optional[Callable] = None):
        """Global top-k from the gathered blocks; merge_fn(scores[S,nq,k], idx[S,nq,k]) replaces the
        HIP merge in CPU tests."""
        if merge_fn is not None:
            s, i = self.blocks()
            return merge_fn(s.contiguous(), i.contiguous())
        from .. import ops
        return ops.topk_merge_packed(self.gathered, self.world, self.nq, self.k, self.idx_offset, self.merged_s, self.merged_i)


class ShardedIndex:
    """This rank's shard of a row-sharded embedding database.

    search_fn(db, q, k, row_offset=..., **kw) -> (scores, idx) and merge_fn(scores[S,nq,k],
    idx[S,nq,k]) -> (scores, idx) default to the HIP entry points (ops.ip_topk / ops.topk_merge);
    tests on CPU inject oracle-backed callables to exercise the sharding and collective logic.
    """

    def __init__(self, shard, row_offset: int, group=None, search_fn: Optional[Callable] = None,
                 merge_fn: Optional[Callable] = None, **search_kwargs):
        if search_fn is None or merge_fn is None:
            from .. import ops
            search_fn = search_fn or ops.ip_topk
            merge_fn = merge_fn or ops.topk_merge
        self.shard = shard
        self.row_offset = int(row_offset)
        self.group = group
        self.search_fn = search_fn
        self.merge_fn = merge_fn
        self.search_kwargs = search_kwargs

    def search_local(self, q, k: int, **kw):
        """Top-k over this rank's rows only; rows are numbered globally."""
        args = dict(self.search_kwargs)
        args.update(kw)
        return self.search_fn(self.shard, q, k, row_offset=self.row_offset, **args)

    def search(self, q, k: int, **kw):
        """Global top-k, identical on every rank: local scan -> one all-gather -> merge."""
        s, i = self.search_local(q, k, **kw)
        gs, gi = allgather_results(s, i, self.group)
        if gs.shape[0] == 1:
            return s, i
        return self.merge_fn(gs, gi)
