"""Diagnostic: where the C5 query's embed latency goes (host preparation, copies, launches, device time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import weights as W
from merizo_search_amd.foldclass.chopping import domains_from_chopping
sd = W.synthetic_state_dict(0)
weights, pe = W.pack_state_dict(sd)
enc = ops.EgnnEncoder(weights, pe, "cuda:0")
pdb = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "AF-Q96PD2-F1-model_v4_ca.pdb")
doms = [d["coords"] for d in domains_from_chopping(pdb, "71-189,190-290,291-453", "A")]
for _ in range(20): enc.embed(doms)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N): enc.embed(doms); torch.cuda.synchronize()
print("embed + sync: %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): enc.embed(doms)
torch.cuda.synchronize()
print("embed back to back (no sync between calls): %.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(50):
    e0.record(); enc.embed(doms); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print("device span of one call (events around it): %.1f us median" % np.median(ts))
