import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn
n, nq, k = (int(x) for x in sys.argv[1].split(","))
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
q_raw = torch.randn(nq, 128, device="cuda") * 3
ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
lib = _lib.load(); lib.ms_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for it in range(3):
    s1, i1 = ops.ip_topk_prefiltered(d, q_raw, k, 1.0, mode=ops.MODE_IP_NORMQ, workspace=ws)
    print("call", it, "fell back:", ops.prefilter_fell_back(ws))
words = 524288 + 8 * 8 * 4096
buf = np.zeros(words, dtype=np.uint64); assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
rec = buf[524288:].reshape(-1, 8, 8)
hits = np.argwhere(rec[:, :4, 0] > 0)
print("waves with a bad tile:", len(hits))
for b, w in hits[:25]:
    o = rec[b, w]
    print(f"  wg {b} wave {w}: tile {int(o[0])-1} of {int(o[1])}, landed {int(o[2])} (seen {int(o[3])}), consumed {[int(o[4] & 0xffffffff)//64, int(o[4]>>32)//64, int(o[5] & 0xffffffff)//64, int(o[5]>>32)//64]}, lanes {int(o[6]):016x}, acc0 {np.array([o[7]], dtype=np.uint64).astype(np.uint32).view(np.float32)[0]:.3e}")
