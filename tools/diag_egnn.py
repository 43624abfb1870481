"""GPU diagnostics for the encoder: error vs goldens, throughput on TED-like batches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import weights as W, synthetic as syn
weights, pe = W.pack_state_dict(W.synthetic_state_dict(0))
enc = ops.EgnnEncoder(weights, pe)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "egnn.npz"))
for c in ["M0", "3w5h", "AF-Q96HM7-F1-model_v4", "AF-Q96PD2-F1-model_v4", "walk1", "walk2", "walk64", "walk257"]:
    e = enc.embed([g[f"coords_{c}"]]).cpu().numpy()[0]; r = g[f"emb_{c}"]
    print(c, "rel err", float(np.abs(e - r).max() / np.abs(r).max()), "cos-1", float(np.dot(e, r) / np.linalg.norm(e) / np.linalg.norm(r) - 1))
for nb in (1, 100, 1000):
    lens = syn.ted_lengths(nb, seed=5) if nb > 1 else np.array([775])
    coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
    for _ in range(2): enc.embed(coords)
    torch.cuda.synchronize(); t = time.time()
    reps = 3
    for _ in range(reps): enc.embed(coords)
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    fl = sum(2 * (263680.0 * n * n + 525312.0 * n) for n in lens.astype(np.float64))
    print(f"nb={nb} sumN={int(lens.sum())} sumN2={int((lens.astype(np.int64)**2).sum())}: {dt*1e3:.2f} ms  {nb/dt:.0f} embeds/s  {fl/dt/1e12:.1f} TFLOP/s alg ({fl/dt/157.3e12*100:.1f}% of fp32 MFMA peak)")
