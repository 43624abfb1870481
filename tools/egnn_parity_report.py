"""Measured parity of the encoder against every reference golden (tests/golden/egnn*.npz), for DESIGN.md: max |delta| / max |e| per case.
usage: [MS_EGNN_SPLIT=0] python tools/egnn_parity_report.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from merizo_search_amd import ops
from merizo_search_amd.foldclass import weights as W
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rows = []
def run(sd, cases):
    w, pe = W.pack_state_dict(sd)
    enc = ops.EgnnEncoder(w, pe, "cuda:0")
    for name, coords, ref in cases:
        e = enc.embed([coords]).cpu().numpy()[0]
        rows.append((name, len(coords), float(np.abs(e - ref).max() / np.abs(ref).max())))
g = np.load(os.path.join(G, "egnn.npz"))
run(W.synthetic_state_dict(0), [(c, g["coords_" + c], g["emb_" + c]) for c in ("M0", "3w5h", "AF-Q96HM7-F1-model_v4", "AF-Q96PD2-F1-model_v4", "walk1", "walk2", "walk64", "walk257")])
gl = np.load(os.path.join(G, "egnn_long.npz"))
run(W.synthetic_state_dict(0), [("std walk%d" % n, gl["coords_walk%d" % n], gl["emb_std_walk%d" % n]) for n in (1000, 2000)])
g2 = np.load(os.path.join(G, "egnn_d2.npz"))
run(W.synthetic_state_dict(0, d2_scale=1.0), [("d2 " + c, g2["coords_" + c], g2["emb_" + c]) for c in ("M0", "walk97", "walk292")] +
    [("d2 walk%d" % n, gl["coords_walk%d" % n], gl["emb_d2_walk%d" % n]) for n in (1000, 2000)])
print("edge GEMM form:", "fp32" if os.environ.get("MS_EGNN_SPLIT", "1") == "0" else "split-bf16 (default)")
for name, n, rel in rows:
    print("  %-28s N=%-5d max|delta|/max|e| = %.2e" % (name, n, rel))
print("  worst: %.2e" % max(r[2] for r in rows))
