"""Diagnostic: the prefiltered search against the plain one (bit equality, time, did the exact pipeline run)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1_000_000, 256, 10), (1_000_000, 256, 1), (4_000_000, 256, 10), (1_000_000, 1024, 10), (1_000_000, 256, 32)]
for n, nq, k in cases:
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    q_raw = torch.randn(nq, 128, device="cuda") * 3
    s0, i0 = ops.ip_topk(d, q_raw, k, mode=ops.MODE_IP_NORMQ)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    s1, i1 = ops.ip_topk_prefiltered(d, q_raw, k, 1.0, mode=ops.MODE_IP_NORMQ, workspace=ws)
    fb = ops.prefilter_fell_back(ws)
    same = torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    res = []
    wsp = ops.TopKWorkspace(d.device).get(n, nq, k)
    out = (torch.empty_like(s0), torch.empty_like(i0))
    for fn in (lambda: ops.ip_topk(d, q_raw, k, mode=ops.MODE_IP_NORMQ, workspace=wsp, out=out),
               lambda: ops.ip_topk_prefiltered(d, q_raw, k, 1.0, mode=ops.MODE_IP_NORMQ, workspace=ws, out=out)):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e3)
    print(f"n={n} nq={nq} k={k}: identical={same} fell_back={fb} | plain {res[0]:.3f} ms, prefiltered {res[1]:.3f} ms ({res[0]/res[1]:.2f}x; {512*n/res[1]/8e9*100:.1f}% of 8 TB/s)", flush=True)
    if not same:
        bad = (i0 != i1).any(dim=1).nonzero().flatten()[:5].tolist()
        print("   first differing queries:", bad, i0[bad[0]].tolist(), i1[bad[0]].tolist(), s0[bad[0]].tolist(), s1[bad[0]].tolist())
    del d
