import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n, nq, k = 1_000_000, 256, 10
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
img = ops.pf_choose_format(d, ops.pf_build_image(d, row_norm_bound=1.0 + 1e-6), 1.0 + 1e-6)
q = torch.randn(nq, 128, device="cuda") * 3
ws = torch.empty_like(ops.PrefilterWorkspace(d.device).get(n, nq, k))
out = (torch.empty(nq, k, device="cuda"), torch.empty(nq, k, dtype=torch.int64, device="cuda"))
kw = dict(row_norm_bound=1.0 + 1e-6, image=img, mode=ops.MODE_IP_NORMQ)
def staged():
    ops.ip_topk_prefiltered_stage("prepare", d, q, k, ws, **kw)
    ops.ip_topk_prefiltered_stage("scan", d, q, k, ws, **kw)
    ops.ip_topk_prefiltered_stage("finish", d, q, k, ws, out=out, **kw)
def one():
    ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, mode=ops.MODE_IP_NORMQ, workspace=ws, out=out, image=img)
for name, f in (("staged", staged), ("one call", one)):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): f()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: host enqueue %.1f us per step, wall %.1f us per step" % (name, t_host / 300 * 1e6, t_all / 300 * 1e6))
# the bench's way: HIP events around the scan launch of every 4th step
for stride in (1, 4, 0):
    for _ in range(50): staged()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(300)]
    t0 = time.perf_counter()
    for s_ in range(300):
        ops.ip_topk_prefiltered_stage("prepare", d, q, k, ws, **kw)
        if stride and s_ % stride == 0: evs[s_][0].record()
        ops.ip_topk_prefiltered_stage("scan", d, q, k, ws, **kw)
        if stride and s_ % stride == 0: evs[s_][1].record()
        ops.ip_topk_prefiltered_stage("finish", d, q, k, ws, out=out, **kw)
    torch.cuda.synchronize()
    print("staged, events every %s steps: wall %.1f us per step" % (stride or "no", (time.perf_counter() - t0) / 300 * 1e6))
# after a burst of fp32 scans (the bench runs its fp32 block first)
ws2 = torch.empty_like(ops.TopKWorkspace(d.device).get(n, nq, k))
for _ in range(200): ops.ip_topk(d, q, k, mode=ops.MODE_IP_NORMQ, workspace=ws2)
torch.cuda.synchronize()
for reps in (20, 200):
    t0 = time.perf_counter()
    for _ in range(reps): staged()
    torch.cuda.synchronize()
    print("staged right after 200 fp32 scans, %d steps: wall %.1f us per step" % (reps, (time.perf_counter() - t0) / reps * 1e6))
