"""One encoder workload for profiling: python tools/prof_egnn.py [n_structures] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import weights as W, synthetic as syn
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
weights, pe = W.pack_state_dict(W.synthetic_state_dict(0))
enc = ops.EgnnEncoder(weights, pe)
lens = syn.ted_lengths(nb, seed=5)
coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
for _ in range(reps):
    enc.embed(coords)
torch.cuda.synchronize()
