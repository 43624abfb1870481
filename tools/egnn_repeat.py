"""Diagnostic: run-to-run bit identity of the encoder (the edge kernel has asynchronous asm loads in flight across
compiler-scheduled matrix instructions): python tools/egnn_repeat.py [reps]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops
from merizo_search_amd.foldclass import weights as W, synthetic as syn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
weights, pe = W.pack_state_dict(W.synthetic_state_dict(0))
enc = ops.EgnnEncoder(weights, pe)
for nb in (3, 200):
    lens = syn.ted_lengths(nb, seed=5)
    coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
    seen = collections.Counter()
    for _ in range(reps):
        e = enc.embed(coords)
        seen[int(e.view(torch.int32).to(torch.int64).sum())] += 1
    print(f"{nb} structures: {len(seen)} distinct result(s) over {reps} runs: {sorted(seen.values(), reverse=True)}")
