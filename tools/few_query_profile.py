"""The few-query regime (1..32 raw queries, one C-ABI call per step) at a CLI-sized database: `python tools/few_query_profile.py [rows] [nq ...]`
under `rocprofv3 --kernel-trace --stats` names the launches of a step over the fp32 rows and over the fp16 image."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nqs = [int(a) for a in sys.argv[2:]] or [1, 8, 32]
k = 10
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
img = ops.pf_build_image(d, row_norm_bound=1.0 + 1e-6).as_format(ops.PF_F16X2)
for nq in nqs:
    q = torch.randn(nq, 128, device="cuda") * 3
    out = (torch.empty(nq, k, device="cuda"), torch.empty(nq, k, dtype=torch.int64, device="cuda"))
    ws = torch.empty_like(ops.TopKWorkspace(d.device).get(n, nq, k))
    wsp = torch.empty_like(ops.PrefilterWorkspace(d.device).get(n, nq, k))
    for name, f in (("fp32 rows", lambda: ops.ip_topk(d, q, k, mode=ops.MODE_IP_NORMQ, workspace=ws, out=out)),
                    ("fp16 image", lambda: ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, mode=ops.MODE_IP_NORMQ, workspace=wsp, out=out, image=img))):
        for _ in range(30): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
        byts = n * (512 if name == "fp32 rows" else 256)
        print("n=%d nq=%d %-10s %.1f us per step = %.3f of 8 TB/s in the bytes it reads" % (n, nq, name, us, byts / us / 8e6))
