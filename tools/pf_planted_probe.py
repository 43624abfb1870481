"""Does a database with planted near-duplicates of the queries (bench.py's: three per query) cost the prefiltered C2 step more than a plain
one?  Per-stage HIP events for {bench db, plain db} x {bench queries, fresh queries}; profiles/r06_hist_step_ab.log holds the answer
before and after the histogram's bucket width was taken from the slope of the sample's tail (ms_sample_bound_kernel)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from merizo_search_amd import ops
from merizo_search_amd.foldclass import sharded
from merizo_search_amd.foldclass import synthetic as syn
dev = torch.device("cuda", 0)
n, nq, k = 1_000_000, 256, 10
b = bench.SearchBench(torch, None, ops, syn, sharded, dev, 0, 1, n, 0, n, nq, k)
bp = b.variant(True)
out = (torch.empty(nq, k, device="cuda"), torch.empty(nq, k, dtype=torch.int64, device="cuda"))
d0 = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
img0 = ops.pf_build_image(d0, row_norm_bound=bp.row_norm_bound).as_format(bp.image.format)
q2 = torch.randn(nq, 128, device="cuda") * 3
def timed(d, img, q, ws, label):
    kw = dict(row_norm_bound=bp.row_norm_bound, image=img, mode=ops.MODE_IP_NORMQ)
    def st(ev=None):
        if ev: ev[0].record()
        ops.ip_topk_prefiltered_stage("prepare", d, q, k, ws, **kw)
        if ev: ev[1].record()
        ops.ip_topk_prefiltered_stage("scan", d, q, k, ws, **kw)
        if ev: ev[2].record()
        ops.ip_topk_prefiltered_stage("finish", d, q, k, ws, out=out, **kw)
        if ev: ev[3].record()
    for _ in range(50): st()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): st()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 300 * 1e6
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(100)]
    for e in evs: st(e)
    torch.cuda.synchronize()
    import numpy as np
    seg = [np.mean([e[i].elapsed_time(e[i + 1]) for e in evs]) * 1e3 for i in range(3)]
    print("%-40s wall %.1f us; prepare %.1f scan %.1f finish %.1f; cands %s flagged %s" % (label, wall, *seg, ops.prefilter_candidates(ws) if hasattr(ops, "prefilter_candidates") else "?", ops.prefilter_flagged(ws)))
timed(bp.db, bp.image, bp.q_raw, bp.ws, "bench db (planted), bench queries")
timed(bp.db, bp.image, q2, bp.ws, "bench db (planted), fresh queries")
timed(d0, img0, bp.q_raw, bp.ws, "plain db, bench queries")
timed(d0, img0, q2, bp.ws, "plain db, fresh queries")
for rep in range(2):
    el, sc, res = b.run(200, 10)
    print("fp32 run (planted db): %.1f us per step, scan %.1f" % (el / 200 * 1e6, sc * 1e3))
