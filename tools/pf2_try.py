"""The prefilter scan over each image arithmetic (f16x2, f16x1, bf16x3) and without an image: a few shapes against the fp32 scan, with
timings.  usage: python tools/pf2_try.py [n,nq,k ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn

def run(n, nq, k, reps=int(os.environ.get("MS_TRY_REPS", "20"))):
    d = syn.device_database(n, 0, seed=0, device="cuda", normalize=True)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    q = torch.randn((nq, 128), generator=g, device="cuda"); q = q / q.norm(dim=1, keepdim=True)
    torch.cuda.synchronize(); t = time.perf_counter(); img = ops.pf_build_image(d, fmt=ops.PF_F16X2, row_norm_bound=1.0 + 1e-6); torch.cuda.synchronize(); t_img = time.perf_counter() - t
    img3 = ops.pf_build_image(d, fmt=ops.PF_BF16X3)
    ws = torch.empty_like(ops.PrefilterWorkspace(d.device).get(n, nq, k))
    s0, i0 = ops.ip_topk(d, q, k)
    res = {}
    only = [f for f in os.environ.get("MS_TRY_FORMATS", "").split(",") if f]          # e.g. MS_TRY_FORMATS=f16x1,f16x2
    for name, image in (("f16x2", img), ("f16x1", img.as_format(ops.PF_F16X1)), ("bf16x3", img3), ("regs", None)):
        if only and name not in only:
            continue
        s, i = ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, workspace=ws, image=image)
        torch.cuda.synchronize()
        ok = bool(torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32)))
        fl = ops.prefilter_flagged(ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts, tt = [], []
        for _ in range(reps):
            ops.ip_topk_prefiltered_stage("prepare", d, q, k, ws, image=image)
            e0.record()
            ops.ip_topk_prefiltered_stage("scan", d, q, k, ws, image=image)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = time.perf_counter()
        for _ in range(reps):
            ops.ip_topk_prefiltered(d, q, k, 1.0 + 1e-6, workspace=ws, image=image, out=(s, i))
        torch.cuda.synchronize()
        res[name] = (ok, fl, float(np.median(ts)), (time.perf_counter() - t) / reps * 1e3)
    print("n=%d nq=%d k=%d fp16 image build %.2f ms\n   " % (n, nq, k, t_img * 1e3) +
          "\n   ".join("%s: identical=%s flagged=%d scan %.3f ms call %.3f ms" % ((nm,) + v) for nm, v in res.items()), flush=True)

shapes = [(70_000, 100, 5), (300_000, 256, 10), (1_000_000, 256, 10), (1_000_000, 128, 10), (4_000_000, 256, 10), (1_000_000, 1024, 10), (1_000_000, 256, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for sh in shapes:
    run(*sh)
