import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
g = torch.Generator(device="cuda")
found = 0
for seed in range(300):
    g.manual_seed(seed)
    n = 100_000 + 37 * seed * 1000 % 400_000; nq = 100 + (seed * 53) % 900; k = 48
    db = torch.randn((n, 128), generator=g, device="cuda"); q = torch.randn((nq, 128), generator=g, device="cuda")
    db = db / db.norm(dim=1, keepdim=True)
    if seed % 2: db = db * (0.1 + 4.0 * torch.rand((n, 1), generator=g, device="cuda"))
    bound = float(1.0 / ops.row_inv_norms(db, 1e-30).min()) * (1 + 1e-6)
    s0, i0 = ops.ip_topk(db, q, k, mode=ops.MODE_IP_NORMQ)
    ws = ops.PrefilterWorkspace(db.device).get(n, nq, k)
    s1, i1 = ops.ip_topk_prefiltered(db, q, k, bound, mode=ops.MODE_IP_NORMQ, workspace=ws)
    fb = ops.prefilter_fell_back(ws)
    if not (torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))):
        found += 1
        badq = ((i0 != i1) | (s0.view(torch.int32) != s1.view(torch.int32))).any(dim=1).nonzero().flatten()
        j = int(badq[0])
        ranks = ((i0[j] != i1[j]) | (s0[j].view(torch.int32) != s1[j].view(torch.int32))).nonzero().flatten().tolist()
        print(f"seed {seed} n={n} nq={nq} fell_back={fb} bad queries {len(badq)} first {j} ranks {ranks[:8]}")
        print("   plain ", i0[j][ranks[0]-1:ranks[0]+3].tolist(), s0[j][ranks[0]-1:ranks[0]+3].tolist())
        print("   pf    ", i1[j][ranks[0]-1:ranks[0]+3].tolist(), s1[j][ranks[0]-1:ranks[0]+3].tolist())
        # is the plain answer right?  brute force in float64 on that query
        qn = (q[j] / q[j].norm()).double()
        sc = db.double() @ qn
        top = torch.topk(sc, k + 2)
        print("   f64   ", top.indices[ranks[0]-1:ranks[0]+3].tolist(), top.values[ranks[0]-1:ranks[0]+3].tolist())
        if found >= 3: break
    del db, q, ws
print("done", found)
