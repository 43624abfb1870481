"""Driver path vs bench step on the same shape: python tools/cli_vs_bench.py [rows] [nq] [k]

Builds a faiss-layout matrix file of `rows` unit rows under /tmp, opens it the way dbsearch_faiss does
(memmap -> engine.upload_rows: ONE resident tensor -> ONE scan per batch through knn_exact), and times a
query batch through (a) the driver's calls and (b) the three-stage calls bench.py times.  Also times the
out-of-core branch (engine.device_blocks: pinned double-buffered H2D per 262,144-row block)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import dbsearch as ds, dbutil, synthetic as syn
from merizo_search_amd.foldclass.engine import HipEngine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = HipEngine("cuda:0")
path = "/tmp/cli_vs_bench_%d.db" % rows
d = syn.device_database(rows, 0, 0, "cuda:0")
d.cpu().numpy().tofile(path)
mm = dbutil.db_memmap(path, (rows, 128))
t = time.perf_counter(); shard = eng.upload_rows(mm, 0, rows); torch.cuda.synchronize(); up = time.perf_counter() - t
assert torch.equal(shard, d)
print("upload_rows, first call (allocates and pins the staging buffers): %.2f s for %.2f GB = %.1f GB/s (page cache -> pinned -> HBM), resident budget %.1f GB"
      % (up, rows * 512 / 1e9, rows * 512 / 1e9 / up, eng.resident_budget(nq, k) / 1e9))
del shard
t = time.perf_counter(); shard = eng.upload_rows(mm, 0, rows); torch.cuda.synchronize(); up = time.perf_counter() - t
assert torch.equal(shard, d)
print("upload_rows, staging buffers in place: %.2f s = %.1f GB/s with %d host threads" % (up, rows * 512 / 1e9 / up, eng.COPY_THREADS))
q = torch.randn(nq, 128, device="cuda"); q = q / q.norm(dim=1, keepdim=True)
import logging
quiet = logging.getLogger("quiet"); quiet.setLevel(logging.ERROR)

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

drv = timeit(lambda: ds.knn_exact(q, [shard], k, eng, log=quiet, to_host=False))
ws = ops.TopKWorkspace(shard.device).get(rows, nq, k)
out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
def staged():
    ops.ip_topk_prepare(shard, q, k, ws); ops.ip_topk_scan(shard, q, k, ws); ops.ip_topk_finish(rows, nq, k, ws, out_s, out_i)
bench = timeit(staged)
print("rows=%d nq=%d k=%d: driver knn_exact (one resident scan) %.3f ms per batch, bench's staged calls %.3f ms (%.1f%% apart)"
      % (rows, nq, k, drv, bench, (drv / bench - 1) * 100))
stream = timeit(lambda: ds.knn_exact(q, dbutil.db_iterator(mm, 262144), k, eng, log=quiet, to_host=False), reps=3)
print("out-of-core branch (blocks of 262,144 rows from the page cache): %.1f ms per batch = %.1f GB/s" % (stream, rows * 512 / 1e6 / stream))
os.remove(path)
