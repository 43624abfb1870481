"""The bench's c3_search workload for profiling (500,000 unit rows, 1000 queries, cosine + length mask, top-10):
python tools/prof_c3.py [reps] [prefiltered]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, nq, mincov, k, dev = 500_000, 1000, 0.7, 10, "cuda:0"
db = syn.device_database(n, 0, seed=3, device=dev, normalize=False) * 2.5
lengths = torch.from_numpy(syn.ted_lengths(n, seed=4).astype(np.float32)).to(dev)
qlen = torch.from_numpy(syn.ted_lengths(nq, seed=5).astype(np.float32)).to(dev)
g = torch.Generator(device=dev); g.manual_seed(6)
q = torch.randn((nq, 128), generator=g, device=dev, dtype=torch.float32)
unit = ops.l2_normalize_rows_(db.clone(), 1e-8)
ws = ops.TopKWorkspace(dev).get(n, nq, k)
out_s = torch.empty((nq, k), dtype=torch.float32, device=dev); out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=lengths, qlen=qlen, mincov=mincov)
if "prefiltered" in sys.argv:
    img = ops.pf_build_image(unit, row_norm_bound=1.0 + 1e-5)
    if ops.pf_format_is_auto():
        img = ops.pf_choose_format(unit, img, 1.0 + 1e-5)
    print("image format:", {0: "bf16x3", 1: "f16x2", 2: "f16x1"}[img.format], flush=True)
    pws = ops.PrefilterWorkspace(dev).get(n, nq, k)
    for _ in range(reps):
        ops.ip_topk_prefiltered(unit, q, k, 1.0 + 1e-5, workspace=pws, image=img, out=(out_s, out_i), **kw)
else:
    for _ in range(reps):
        ops.ip_topk_prepare(unit, q, k, ws, **kw); ops.ip_topk_scan(unit, q, k, ws, **kw); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
torch.cuda.synchronize()
