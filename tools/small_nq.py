"""Diagnostic: step time of the one-call path (MS_MODE_IP_NORMQ) against the staged calls, few queries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
k = 10
for n in (1_000_000, 4_000_000):
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    for nq in (1, 4, 8, 32, 64):
        q_raw = torch.randn(nq, 128, device="cuda") * 3
        q = torch.empty_like(q_raw)
        ws = ops.TopKWorkspace(d.device).get(n, nq, k)
        out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
        def staged():
            ops.l2_normalize_rows(q_raw, 1e-12, out=q); ops.ip_topk_prepare(d, q, k, ws); ops.ip_topk_scan(d, q, k, ws); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
        def onecall():
            ops.ip_topk(d, q_raw, k, mode=ops.MODE_IP_NORMQ, workspace=ws, out=(out_s, out_i))
        res = []
        for fn in (staged, onecall):
            for _ in range(30): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 100 * 1e6)
        print(f"rows={n} nq={nq}: staged calls {res[0]:.1f} us ({512*n/res[0]/8e6*100:.1f}% of 8 TB/s), one call {res[1]:.1f} us ({512*n/res[1]/8e6*100:.1f}%)", flush=True)
    del d
