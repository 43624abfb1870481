"""Diagnostic: encoder throughput on 1000 TED-length domains + the C5 query, as bench.py's embed entry measures them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import synthetic as syn, weights as W
from merizo_search_amd.foldclass.chopping import domains_from_chopping
sd = W.synthetic_state_dict(0)
weights, pe = W.pack_state_dict(sd)
enc = ops.EgnnEncoder(weights, pe, "cuda:0")
lens = syn.ted_lengths(1000, seed=5)
coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
flops = float(sum(2.0 * (263680.0 * n * n + 525312.0 * n) for n in lens.astype(np.float64)))
def timed(batch, reps):
    enc.embed(batch); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); enc.embed(batch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))
t = timed(coords, 5)
print(f"1000 domains: {t*1e3:.1f} ms = {1000/t:.0f} embeds/s = {flops/t/157.3e12*100:.1f}% of fp32 MFMA peak", flush=True)
pdb = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "AF-Q96PD2-F1-model_v4_ca.pdb")
doms = domains_from_chopping(pdb, "71-189,190-290,291-453", "A")
t3 = timed([d["coords"] for d in doms], 30)
whole = np.concatenate([d["coords"] for d in domains_from_chopping(pdb, "1-775", "A")])
tw = timed([whole], 10)
print(f"C5 query (3 domains): {t3*1e3:.3f} ms; whole chain N={len(whole)}: {tw*1e3:.2f} ms = {2.0*(263680.0*len(whole)**2+525312.0*len(whole))/tw/157.3e12*100:.1f}%")
