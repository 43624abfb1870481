"""Diagnostic: one (rows, queries) shape of the one-call search in a loop, for rocprofv3 --kernel-trace --stats.
usage: python3 tools/hbm_shape.py ROWS NQ [ITERS]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn
n, nq = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
k = 10
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
q_raw = torch.randn(nq, 128, device="cuda") * 3
ws = ops.TopKWorkspace(d.device).get(n, nq, k)
out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
for _ in range(30): ops.ip_topk(d, q_raw, k, mode=ops.MODE_IP_NORMQ, workspace=ws, out=(out_s, out_i))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): ops.ip_topk(d, q_raw, k, mode=ops.MODE_IP_NORMQ, workspace=ws, out=(out_s, out_i))
torch.cuda.synchronize(); us = (time.perf_counter() - t0) / iters * 1e6
print(f"rows={n} nq={nq}: {us:.1f} us/search = {512*n/us/8e6*100:.1f}% of 8 TB/s")
