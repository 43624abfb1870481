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

The drivers (dbsearch.py, makedb.py) and the CLI use this module whenever a process group is
initialised -- `python -m torch.distributed.run --nproc-per-node N -m merizo_search_amd.cli
search ...` (cli.init_distributed): every rank uploads only its `shard_bounds` rows, the query
structures are embedded data-parallel (`embed_distributed`, split by sum N^2), and rank 0 alone
retrieves hit records and writes the TSV files.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

BACKEND_ENV = "MERIZO_DIST_BACKEND"          # "nccl" (= RCCL, default) | "gloo" (self-tests on a one-GPU box)
SAME_DEVICE_ENV = "MERIZO_SAME_DEVICE"       # "1": every rank uses cuda:0 (self-tests on a one-GPU box; gloo only)
TIMEOUT_ENV = "MERIZO_DIST_TIMEOUT_S"        # collective timeout of the process group in seconds (default: one day)
FINISH_TIMEOUT_ENV = "MERIZO_FINISH_TIMEOUT_S"   # how long ranks != 0 wait for rank 0's post-processing (default: 30 days)
_FINISH_KEY, _ACK_KEY = "merizo_search_amd/finished", "merizo_search_amd/acks"


def _env_seconds(name: str, default: float):
    from datetime import timedelta

    try:
        v = float(os.environ.get(name, default))
    except ValueError:
        v = default
    return timedelta(seconds=max(v, 1.0))


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world size) of the initialised process group, (0, 1) without one."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def init_distributed() -> Tuple[int, int, Optional[str]]:
    """Join the process group torchrun describes in the environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*).  -> (rank, world, device name of this rank or None when WORLD_SIZE is 1).

    Must run before anything else touches the GPU in this process; nothing is exec'd afterwards.
    Backend "nccl" is RCCL on ROCm (xGMI between the GPUs of a node)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        rank = dist.get_rank()
        local = int(os.environ.get("LOCAL_RANK", rank))
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    same_device = os.environ.get(SAME_DEVICE_ENV) == "1"
    backend = os.environ.get(BACKEND_ENV, "nccl")
    if same_device and backend == "nccl" and not dist.is_initialized():
        # RCCL needs one GPU per rank: ranks that share a device fail or hang inside the communicator set-up
        raise RuntimeError(f"{SAME_DEVICE_ENV}=1 puts every rank on cuda:0, which the nccl (RCCL) backend cannot do: "
                           f"set {BACKEND_ENV}=gloo for one-GPU self-tests")
    index = 0 if same_device else local
    device = torch.device("cuda", index)
    if torch.cuda.is_available():          # (without a GPU the engine set-up that follows fails loudly)
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        # Collectives here are short (a few hundred KB per batch), but rank 0's serial post-processing (TM-align per hit,
        # the multi-domain step, the TSV files) is not: the other ranks must never sit in a device collective with the
        # default 10-minute watchdog while it runs (finalize_distributed waits on the store instead), and the group's own
        # timeout is generous.
        timeout = _env_seconds(TIMEOUT_ENV, 86400.0)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=timeout)
        else:
            dist.init_process_group(backend, timeout=timeout)
    return dist.get_rank(), dist.get_world_size(), f"cuda:{index}"


def finalize_distributed() -> None:
    """End of a multi-rank run.  Ranks other than 0 return from the drivers right after the exchange, while rank 0 still
    runs its serial post-processing, possibly for hours.  They therefore do NOT wait in a collective (a pending RCCL
    barrier keeps their GPUs spinning and trips the watchdog after the group's timeout): rank 0 sets a key in the process
    group's store when it is done, the others block on that key (a host-side socket wait), acknowledge, and everybody
    tears the group down; rank 0 waits for the acknowledgements so that the store outlives its readers."""
    import time

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    store = None
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:                       # (private accessor: fall back to a collective if a torch release drops it)
        store = None
    if world > 1 and store is not None:
        limit = _env_seconds(FINISH_TIMEOUT_ENV, 30 * 86400.0)
        if rank == 0:
            store.set(_FINISH_KEY, "1")
            t0 = time.monotonic()
            while int(store.add(_ACK_KEY, 0)) < world - 1 and time.monotonic() - t0 < 600.0:
                time.sleep(0.01)
        else:
            store.wait([_FINISH_KEY], limit)
            store.add(_ACK_KEY, 1)
    elif world > 1:
        dist.barrier()
    dist.destroy_process_group()


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


def exchange_and_merge(scores, idx, engine, group=None):
    """Per-shard results (float32 [nq,k], int64 [nq,k], rows numbered globally) -> the global top-k,
    identical on every rank: one all-gather of the packed blocks + one merge.  No-op at world size 1.
    `engine.merge_gathered(exchange)` runs the merge (HIP: ms_topk_merge_strided on the gathered
    buffer in place)."""
    _rank, world = rank_world(group)
    if world == 1:
        return scores, idx
    nq, k = scores.shape
    ex = PackedExchange(nq, k, scores.device, group)
    ex.out_s.copy_(scores)
    ex.out_i.copy_(idx)
    ex.exchange()
    return engine.merge_gathered(ex)


def balance_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Deterministic longest-processing-time split of item numbers over `world` ranks (every rank
    computes the same answer).  Items of a rank are in ascending order."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    mine: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        mine[r].append(i)
        load[r] += float(costs[i])
    return [sorted(m) for m in mine]


def embed_distributed(network, coords_list: Sequence[np.ndarray], group=None):
    """Embed structures data-parallel over the ranks (no exchange inside the encoder): rank r embeds
    its share of the ragged batch, balanced by sum N^2 (the encoder's cost), and ONE all-gather of
    float32 [max share, 128] blocks gives every rank all embeddings, in input order.  A ragged batch
    and one-by-one launches are bit-identical (tests/test_egnn_gpu.py), so the split does not change
    any embedding."""
    rank, world = rank_world(group)
    if world == 1 or len(coords_list) < 2:
        return network.embed_many(coords_list)
    import torch
    import torch.distributed as dist

    shares = balance_by_cost([float(np.asarray(c).shape[0]) ** 2 for c in coords_list], world)
    cap = max(len(s) for s in shares)
    mine = shares[rank]
    local = network.embed_many([coords_list[i] for i in mine]) if mine else None
    device = local.device if local is not None else network.engine.device
    block = torch.zeros((cap, 128), dtype=torch.float32, device=device)
    if mine:
        block[: len(mine)] = local
    gathered = torch.empty((world, cap, 128), dtype=torch.float32, device=device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(gathered, block, group=group)
    else:
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block, group=group)
        gathered.copy_(torch.stack(parts))
    out = torch.empty((len(coords_list), 128), dtype=torch.float32, device=device)
    for r, share in enumerate(shares):
        if share:
            out[torch.as_tensor(share, device=device)] = gathered[r, : len(share)]
    return out


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
