"""Parity of the search path at extreme shapes (many queries, tiny databases, k = n): python tools/extreme_shapes.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
from oracle import oracle as orc
for n, nq, k in [(1000, 20000, 10), (3, 5000, 3), (200000, 3000, 1), (50, 4097, 50), (70000, 1025, 20), (33, 1, 33), (1_500_000, 129, 10)]:
    db = syn.normalized_database(n, 11); q = syn.normalized_database(nq, 12)
    s, i = ops.ip_topk(torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda(), k)
    sr, ir = orc.ip_topk(db, q, k, order=1)
    ok = np.array_equal(i.cpu().numpy(), ir) and np.array_equal(s.cpu().numpy().view(np.uint32), sr.view(np.uint32))
    print(n, nq, k, "OK" if ok else "MISMATCH")
