"""The prefilter's scan launch alone (prepare once, then the scan in a loop under HIP events; no merge, no re-scoring): for diagnostic
builds whose results are not meant to be right (MS_LIB_OVERRIDE=.../build/nowrite/...).
usage: [MS_PF_FORMAT=f16x2|f16x1|bf16x3] python tools/pf_scan_only.py n,nq,k [...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
for a in sys.argv[1:]:
    n, nq, k = (int(x) for x in a.split(","))
    d = syn.device_database(n, 0, seed=0, device="cuda", normalize=True)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    q = torch.randn((nq, 128), generator=g, device="cuda"); q = q / q.norm(dim=1, keepdim=True)
    img = ops.pf_build_image(d, row_norm_bound=1.0 + 1e-6)
    ws = torch.empty_like(ops.PrefilterWorkspace(d.device).get(n, nq, k))
    ops.ip_topk_prefiltered_stage("prepare", d, q, k, ws, image=img)
    for _ in range(20):
        ops.ip_topk_prefiltered_stage("scan", d, q, k, ws, image=img)
    torch.cuda.synchronize()
    ts = []
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.ip_topk_prefiltered_stage("scan", d, q, k, ws, image=img); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("n=%d nq=%d k=%d format=%s: scan launch median %.1f us min %.1f us" % (n, nq, k, os.environ.get("MS_PF_FORMAT", "f16x2"), np.median(ts) * 1e3, np.min(ts) * 1e3), flush=True)
    del d, img, ws
