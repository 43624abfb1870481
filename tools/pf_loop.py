"""One prefiltered search shape in a loop (for rocprofv3): python3 tools/pf_loop.py ROWS NQ K [ITERS] [noimage]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
img = None if "noimage" in sys.argv else ops.pf_build_image(d, row_norm_bound=1.0 + 1e-6)
if img is not None and ops.pf_format_is_auto():
    img = ops.pf_choose_format(d, img, 1.0 + 1e-6)           # (as the engine does: F16X1 or F16X2 over the same image)
print("image format:", None if img is None else {0: "bf16x3", 1: "f16x2", 2: "f16x1"}[img.format], flush=True)
q_raw = torch.randn(nq, 128, device="cuda") * 3
ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
out = (torch.empty(nq, k, device="cuda"), torch.empty(nq, k, dtype=torch.int64, device="cuda"))
kw = dict(mode=ops.MODE_IP_NORMQ, workspace=ws, out=out, image=img)
for _ in range(3 if n > 20_000_000 else 30): ops.ip_topk_prefiltered(d, q_raw, k, 1.0 + 1e-6, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): ops.ip_topk_prefiltered(d, q_raw, k, 1.0 + 1e-6, **kw)
torch.cuda.synchronize(); print(f"n={n} nq={nq} k={k}: {(time.perf_counter()-t0)/iters*1e3:.4f} ms per search, exact-pass queries: {ops.prefilter_flagged(ws)}")
