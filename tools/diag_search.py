"""GPU diagnostics: which k order the fp32 MFMA chain follows, and a first timing of the scan."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn
from oracle import oracle as orc

db = syn.normalized_database(4096, 1); q = syn.normalized_database(64, 2)
s, i = ops.ip_topk(torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda(), 10)
s = s.cpu().numpy(); i = i.cpu().numpy()
for order in (0, 1):
    sr, ir = orc.ip_topk(db, q, 10, order=order)
    print("order", order, "idx equal", np.array_equal(i, ir), "score bits equal", np.array_equal(s.view(np.uint32), sr.view(np.uint32)),
          "max abs diff", float(np.abs(s - sr).max()))

cases = ((1_000_000, 256, 10), (1_000_000, 32, 10), (1_000_000, 1, 10), (4_000_000, 256, 10), (1_000_000, 1024, 10))
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
aux = os.environ.get('DIAG_COSINE') == '1'
for n, nq, k in cases:
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=not aux)
    kw = {}
    if aux:
        kw = dict(mode=ops.MODE_COSINE_RAW, inv_norm=ops.row_inv_norms(d), lengths=torch.from_numpy(syn.ted_lengths(n, 3).astype(np.float32)).cuda(),
                  qlen=torch.from_numpy(syn.ted_lengths(nq, 4).astype(np.float32)).cuda(), mincov=0.7)
    qq = torch.randn(nq, 128, device="cuda"); qq = qq / qq.norm(dim=1, keepdim=True)
    w = ops.TopKWorkspace(d.device); ws = w.get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.ip_topk_prepare(d, qq, k, ws, **kw); ops.ip_topk_scan(d, qq, k, ws, **kw); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = int(os.environ.get('DIAG_REPS', '30')); tsl = []; tml = []
    for _ in range(reps):
        e[0].record(); ops.ip_topk_prepare(d, qq, k, ws, **kw); ops.ip_topk_scan(d, qq, k, ws, **kw); e[1].record(); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i); e[2].record()
        torch.cuda.synchronize(); tsl.append(e[0].elapsed_time(e[1])); tml.append(e[1].elapsed_time(e[2]))
    ts = float(np.median(tsl)); tm = float(np.median(tml))
    fl = 2.0 * 128 * nq * n
    print(f"n={n} nq={nq} k={k}: scan {ts*1e3:.1f} us merge {tm*1e3:.1f} us | {fl/ts/1e9:.1f} TFLOP/s ({fl/ts/1e9/157.3*100:.1f}% mfma) "
          f"{n*512/ts/1e9*1e3/1e3:.2f} TB/s ({n*512/ts/1e6/8000*100:.1f}% hbm) | {nq/((ts+tm)/1e3):.0f} q/s")
    del d
