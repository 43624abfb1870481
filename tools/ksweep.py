"""Diagnostic: step time of the C2 shape at several k, and the C3 search half (cosine + mask)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n, nq = 1_000_000, 256
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
q = torch.randn(nq, 128, device="cuda"); q = q / q.norm(dim=1, keepdim=True)
for k in [int(a) for a in sys.argv[1:]] or [1, 10, 20, 32, 64]:
    ws = ops.TopKWorkspace(d.device).get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    def step(ev=None):
        ops.ip_topk_prepare(d, q, k, ws)
        if ev: ev[0].record()
        ops.ip_topk_scan(d, q, k, ws)
        if ev: ev[1].record()
        ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
    for _ in range(60): step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    t0 = time.perf_counter()
    for e in evs: step(e)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 60 * 1e3
    sc = np.mean([a.elapsed_time(b) for a, b in evs])
    print(f"k={k}: {ms:.4f} ms per step, scan {sc * 1e3:.1f} us = {2 * 128 * nq * n / (sc * 1e-3) / 157.3e12 * 100:.1f}% of fp32 MFMA peak", flush=True)
# C3 search half
n, nq, mincov, k = 500_000, 1000, 0.7, 10
db = syn.device_database(n, 0, seed=3, device="cuda:0", normalize=False) * 2.5
lengths = torch.from_numpy(syn.ted_lengths(n, seed=4).astype(np.float32)).cuda()
qlen = torch.from_numpy(syn.ted_lengths(nq, seed=5).astype(np.float32)).cuda()
q = torch.randn(nq, 128, device="cuda")
rows = ops.l2_normalize_rows_(db.clone(), 1e-8)
ws = ops.TopKWorkspace(db.device).get(n, nq, k)
out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=lengths, qlen=qlen, mincov=mincov)
def step(ev=None):
    ops.ip_topk_prepare(rows, q, k, ws, **kw)
    if ev: ev[0].record()
    ops.ip_topk_scan(rows, q, k, ws, **kw)
    if ev: ev[1].record()
    ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
for _ in range(40): step()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
t0 = time.perf_counter()
for e in evs: step(e)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 40 * 1e3
sc = np.mean([a.elapsed_time(b) for a, b in evs])
print(f"c3 search (cosine + mask, 500k x 1000): {ms:.4f} ms per step, scan {sc * 1e3:.1f} us = {2 * 128 * nq * n / (sc * 1e-3) / 157.3e12 * 100:.1f}% of fp32 MFMA peak")
