"""Diagnostic (-DMS_STAMP builds): cycles of the edge kernel's prologue / 13 stages / epilogue per 128-edge tile."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import weights as W, synthetic as syn
lib = _lib.load()
lib.ms_debug_egnn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
weights, pe = W.pack_state_dict(W.synthetic_state_dict(0))
enc = ops.EgnnEncoder(weights, pe)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lens = syn.ted_lengths(nb, seed=5)
coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
for _ in range(3): enc.embed(coords)
torch.cuda.synchronize()
buf = np.zeros(8 * 32768, dtype=np.uint64)
assert lib.ms_debug_egnn_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(-1, 8); st = st[st[:, 5] > 0].astype(np.float64)
m = np.median(st, axis=0)
print(f"tiles {len(st)} (wave 0 of each): per stage  H-phase {m[0]/13:.0f}  barrier {m[1]/13:.0f}  MFMA phase {m[2]/13:.0f}  barrier {m[3]/13:.0f}  | epilogue {m[4]:.0f} cycles (160 MFMA = 10240 nominal)")
