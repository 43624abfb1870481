"""One scan workload for profiling: python tools/prof_scan.py [rows] [nq] [k] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
d = syn.device_database(n, 0, 0, "cuda:0")
q = torch.randn(nq, 128, device="cuda"); q = q / q.norm(dim=1, keepdim=True)
ws = ops.TopKWorkspace(d.device).get(n, nq, k)
out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
for _ in range(reps):
    ops.ip_topk_prepare(d, q, k, ws); ops.ip_topk_scan(d, q, k, ws); ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
torch.cuda.synchronize()
