"""Diagnostic: run-to-run consistency of ms_ip_topk at k = 48 (the loader-wave form with 32-entry lists) on shapes that showed
a rare lost row before the shared bound's counters were read with compiler-visible loads."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merizo_search_amd import ops
g = torch.Generator(device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for seed, n, nq in ((2, 174000, 206), (26, 262000, 578), (46, 202000, 738)):
    g.manual_seed(seed)
    db = torch.randn((n, 128), generator=g, device="cuda"); q = torch.randn((nq, 128), generator=g, device="cuda")
    db = db / db.norm(dim=1, keepdim=True)
    if seed % 2: db = db * (0.1 + 4.0 * torch.rand((n, 1), generator=g, device="cuda"))
    for k in (48, 10):
        seen = collections.Counter()
        for rep in range(reps):
            s0, i0 = ops.ip_topk(db, q, k, mode=ops.MODE_IP_NORMQ)
            seen[(int(i0.sum()), int(s0.view(torch.int32).sum()))] += 1
        print(f"seed {seed} k={k}: {len(seen)} distinct result(s) over {reps} runs: {sorted(seen.values(), reverse=True)}")
