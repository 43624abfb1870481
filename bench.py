#!/usr/bin/env python
"""Foldclass search benchmark: queries/sec of the exact 128-d cosine top-k scan on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1] (C2): brute-force cosine top-10 over a 1M x 128
float32 synthetic database, batch = 256 queries.  A step is one pass of the hot path over one
query batch: normalised queries -> fused Q.D^T + top-k scan of the resident shard -> merge of
the per-chunk lists [-> RCCL all-gather of per-shard top-k + shard merge when N > 1].
With N > 1 the SAME database is row-sharded over the ranks (strong scaling: total work fixed).
The database and the queries are resident in HBM before the timed region starts.
`--rows 365000000 --nq 4096` runs the TED-scale shape (C4).  `--streams 2` keeps two query batches
in flight on two HIP streams (batch i+1's short kernels fill the tail of batch i's scan: +5 % q/s
at C2); the default is one, so that the HIP-event duration of the scan launch is that kernel alone.

One JSON line is printed by rank 0 (contract in the task statement), with
  roofline     for the dominant kernel (ms_scan_loader_kernel; ms_scan_kernel for < 3 query tiles):
               algorithmic flops (2*128*nq*rows per launch) over the HIP-event duration of the
               scan stage, against the fp32 MFMA peak (157.3 TFLOP/s) when nq >= 39, else
               algorithmic bytes (512 B per row) against the 8 TB/s HBM peak; both fractions
               are always included;
  cpu_baseline the CPU oracle (oracle/oracle.c, a restatement of the reference's faiss path)
               timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_F32_PEAK = 157.3e12     # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
HBM_PEAK = 8.0e12            # same guide, "HBM3E peak BW" (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--rows", type=int, default=1_000_000, help="total database rows (sharded over the ranks)")
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="after the timed region, also time the same steps with two batches in flight (reported as `pipelined`)")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="run the multi-GPU exchange + shard merge even on one GPU (validates that code path; slower)")
    ap.add_argument("--streams", type=int, default=1,
                    help="query batches in flight: consecutive steps alternate over this many HIP streams, each with its own "
                         "workspace (batch i+1's short kernels fill the tail of batch i's scan)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    from merizo_search_amd import _lib, ops
    from merizo_search_amd.foldclass import synthetic as syn
    from merizo_search_amd.foldclass.sharded import PackedExchange, allgather_results, shard_bounds

    _lib.require_gpu()                                   # fails loudly without the HIP library / a GPU
    # self-test hooks (one-GPU boxes): MS_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # MS_BENCH_BACKEND=gloo swaps the collective backend, so that the multi-rank logic can be run end to end
    dev_index = 0 if os.environ.get("MS_BENCH_SAME_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MS_BENCH_BACKEND", "nccl")             # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n_total, nq, k = args.rows, args.nq, args.k
    lo, hi = shard_bounds(n_total, world, rank)
    n_local = hi - lo

    # ---- synthetic inputs, resident in HBM: unit rows ~ N(0,1)/|.|, seed 0;
    # queries seed 1; 3 planted near-duplicates per query at known global rows.
    db = syn.device_database(n_local, lo, seed=0, device=dev, normalize=True)
    g = torch.Generator(device=dev); g.manual_seed(1)
    q = torch.randn((nq, 128), generator=g, device=dev, dtype=torch.float32)
    q = q / q.norm(dim=1, keepdim=True)
    gp = torch.Generator(device="cpu"); gp.manual_seed(2)
    planted_rows = torch.randperm(n_total, generator=gp)[: nq * 3].reshape(nq, 3)
    noise = torch.randn((nq, 3, 128), generator=gp) * 0.02
    planted = q.cpu()[:, None, :] + noise
    planted = planted / planted.norm(dim=2, keepdim=True)
    flat_rows = planted_rows.reshape(-1)
    mine = (flat_rows >= lo) & (flat_rows < hi)
    if mine.any():
        db[(flat_rows[mine] - lo).to(dev)] = planted.reshape(-1, 128)[mine].to(dev)

    n_pipes = max(1, args.streams)
    pipes = []
    for _ in range(n_pipes):
        ex = PackedExchange(nq, k, dev)    # this rank's results are written straight into its all-gather block
        pipes.append({"ws": torch.empty_like(ops.TopKWorkspace(dev).get(n_local, nq, k)), "ex": ex,
                      "stream": torch.cuda.Stream(device=dev) if n_pipes > 1 else torch.cuda.current_stream(dev)})
    shard_merge = ops.topk_merge
    step_no = [0]

    def step(events=None):
        pipe = pipes[step_no[0] % n_pipes]
        step_no[0] += 1
        ws, ex = pipe["ws"], pipe["ex"]
        with torch.cuda.stream(pipe["stream"]):
            ops.ip_topk_prepare(db, q, k, ws)                               # queries + sample pass (lower bound)
            if events is not None:
                events[0].record()
            ops.ip_topk_scan(db, q, k, ws)                                  # dominant kernel: ONE scan launch
            if events is not None:
                events[1].record()
            ops.ip_topk_finish(n_local, nq, k, ws, ex.out_s, ex.out_i, row_offset=lo)
            if world > 1 or args.exercise_exchange:
                ex.exchange()                                               # ONE RCCL all-gather of 12*nq*k bytes per rank
                return ex.merge()                                           # merge of the S blocks in place
            return ex.out_s, ex.out_i

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(40):          # untimed preparation like the data generation above: the GPU leaves its idle clocks
        step()                   # within the first ~20 launches, whatever --warmup is
    for _ in range(args.warmup):
        res = step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        res = step(evs[s])
    fence()
    elapsed = time.perf_counter() - t0
    scan_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if world > 1:
        t = torch.tensor([elapsed, scan_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, scan_ms = float(t[0]), float(t[1])

    # ---- after the timed region (not part of `value`): the same steps with two batches in flight
    pipelined = None
    if world == 1 and n_pipes == 1 and args.pipelined:
        extra = {"ws": torch.empty_like(pipes[0]["ws"]), "ex": PackedExchange(nq, k, dev), "stream": torch.cuda.Stream(device=dev)}
        first = dict(pipes[0], stream=torch.cuda.Stream(device=dev))
        saved, pipes[:] = list(pipes), [first, extra]
        n_pipes = 2
        psteps = max(20, min(args.steps, 100))
        for _ in range(10):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(psteps):
            step()
        fence()
        dt = time.perf_counter() - t1
        pipelined = {"batches_in_flight": 2, "steps": psteps, "value": nq * psteps / dt, "unit": "queries/s",
                     "ms_per_step": dt / psteps * 1e3,
                     "note": "same steps alternating over two HIP streams with their own workspaces, timed after the main region"}
        pipes[:] = saved
        n_pipes = 1

    # ---- correctness of what was timed: planted rows recalled, exact top-k on a query sample
    fs, fi = res
    fi_c = fi.cpu()
    recall = float(np.mean([len(set(planted_rows[j].tolist()) & set(fi_c[j].tolist())) / 3.0 for j in range(nq)]))
    sample = min(nq, 8)
    ref = (q[:sample] @ db.T).topk(min(k, n_local), dim=1)
    if world == 1:
        exact = bool(torch.equal(ref.indices + lo, fi[:sample]))
    else:
        gs, gi = allgather_results(ref.values.contiguous(), (ref.indices + lo).contiguous())
        ms_, mi_ = shard_merge(gs, gi)
        exact = bool(torch.equal(mi_, fi[:sample]))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = nq * args.steps / elapsed
        scan_rows = n_local                                                     # the timed launch scans every row of the shard
        flops = 2.0 * 128 * nq * scan_rows
        bytes_ = 512.0 * scan_rows
        t_scan = scan_ms * 1e-3
        mfma_frac = flops / t_scan / MFMA_F32_PEAK
        hbm_frac = bytes_ / t_scan / HBM_PEAK
        if nq >= 39:      # fp32-MFMA-bound above ~39 queries per pass
            roof = {"bound": "mfma", "achieved": flops / t_scan / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": mfma_frac, "traffic": None}
        else:
            roof = {"bound": "hbm", "achieved": bytes_ / t_scan / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": hbm_frac, "traffic": None}
        roof.update({"kernel": "ms_scan_loader_kernel" if (nq > 64 and k <= 20) else "ms_scan_kernel", "kernel_ms": scan_ms,
                     "mfma_frac": mfma_frac, "hbm_frac": hbm_frac, "rows_per_launch": scan_rows,
                     "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bytes_})
        # HBM traffic of one launch of that kernel from the committed PMC passes of this same command
        # (profiles/: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 2x read correction applied)
        pmc = os.path.join(REPO, "profiles", "r01_c2_pmc_v9.json")
        if world == 1 and (n_total, nq, k) == (1_000_000, 256, 10) and os.path.exists(pmc):
            with open(pmc) as fh:
                roof["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
            roof["traffic_source"] = "profiles/r01_c2_pmc_v9.json"
        line = {
            "metric": "queries/sec (exact 128-d cosine top-k, recall@k vs brute force = %.3f)" % recall,
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C2: brute-force cosine top-%d, %d x 128 fp32 synthetic DB, batch=%d queries" % (k, n_total, nq))
                       if (n_total, nq) == (1_000_000, 256) else ("cosine top-%d, %d x 128 fp32 synthetic DB, batch=%d queries" % (k, n_total, nq)),
                       "db_rows": n_total, "dim": 128, "queries_per_step": nq, "k": k, "score": "inner product of unit rows",
                       "sharding": "contiguous row shards, %d rows/GPU, RCCL all-gather of per-shard top-k" % n_local if world > 1 else "single shard"},
            "recall_at_k": recall, "topk_exact_on_sample": exact, "batches_in_flight": n_pipes, "pipelined": pipelined,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(db, q, k, n_total)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(db, q, k, n_total):
    """The CPU oracle's faiss-path restatement (oracle.c:orc_ip_topk, OpenMP + AVX2 FMA) on the
    host cores, on the first `sample` rows of the same database and the same query batch."""
    from oracle import oracle as orc
    cores = orc.num_threads()
    qh = q.cpu().numpy()
    sample = min(db.shape[0], 50_000)
    dbh = db[:sample].cpu().numpy()
    t = time.perf_counter(); orc.ip_topk(dbh, qh, k); dt = time.perf_counter() - t
    # aim for ~5 s of wall time on the final sample
    sample2 = int(min(db.shape[0], max(sample, sample * 5.0 / max(dt, 1e-3))))
    if sample2 > sample:
        dbh = db[:sample2].cpu().numpy()
        t = time.perf_counter(); orc.ip_topk(dbh, qh, k); dt = time.perf_counter() - t
        sample = sample2
    value = qh.shape[0] / (dt * n_total / sample)
    return {"value": value, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": "first %d of %d rows x all %d queries, %.2f s wall, scaled linearly in rows" % (sample, n_total, qh.shape[0], dt)}


if __name__ == "__main__":
    main()
