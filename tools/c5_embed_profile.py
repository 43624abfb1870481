"""The C5 query's embedding (AF-Q96PD2, chopping 71-189,190-290,291-453: three domains) call by call: wall time per call with a
synchronisation after each (what bench.py's c5_query reports), wall time of back-to-back calls, and -- under `rocprofv3 --kernel-trace
--stats` -- the launches of one call."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import weights as W
from merizo_search_amd.foldclass.chopping import domains_from_chopping
weights, pe = W.pack_state_dict(W.synthetic_state_dict(0))
enc = ops.EgnnEncoder(weights, pe, "cuda:0")
doms = domains_from_chopping(os.path.join(R, "tests", "golden", "AF-Q96PD2-F1-model_v4_ca.pdb"), "71-189,190-290,291-453", "A")
batch = [d["coords"] for d in doms]
for _ in range(20): enc.embed(batch)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t = time.perf_counter(); enc.embed(batch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("synchronised after every call: median %.1f us, min %.1f us" % (np.median(ts) * 1e6, min(ts) * 1e6))
t = time.perf_counter()
for _ in range(200): enc.embed(batch)
th = time.perf_counter() - t
torch.cuda.synchronize()
print("back to back: host %.1f us per call, wall %.1f us per call" % (th / 200 * 1e6, (time.perf_counter() - t) / 200 * 1e6))
