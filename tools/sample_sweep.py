"""Diagnostic: step time (prepare + scan + finish) of the fp32 search and of the prefiltered search over the split image for a few
shapes; run once per value of MS_SAMPLE_COEF / MS_PREPASS_TILES (environment, read by the library at load).
usage: python3 tools/sample_sweep.py N,NQ,K [...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1_000_000, 256, 10)]
for n, nq, k in cases:
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    q = torch.randn(nq, 128, device="cuda"); q = q / q.norm(dim=1, keepdim=True)
    ws = ops.TopKWorkspace(d.device).get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    def step():
        ops.ip_topk_prepare(d, q, k, ws); ops.ip_topk_scan(d, q, k, ws); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
    iters = max(10, min(200, int(2e8 / (n * max(nq, 64) / 256))))
    for _ in range(max(5, iters // 4)): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / iters * 1e3
    line = f"n={n} nq={nq} k={k}: fp32 {ms:.4f} ms"
    if ops.prefilter_serves(n, nq, k):
        img = ops.pf_build_image(d)
        pws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
        def pstep(): ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-5, workspace=pws, out=(out_s, out_i), image=img)
        for _ in range(max(5, iters // 4)): pstep()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): pstep()
        torch.cuda.synchronize(); pms = (time.perf_counter() - t0) / iters * 1e3
        line += f" | prefiltered {pms:.4f} ms (flagged {ops.prefilter_flagged(pws)})"
        del img
    print(line, flush=True)
    del d
