import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn
n, nq, k = (int(x) for x in sys.argv[1].split(","))
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
q_raw = torch.randn(nq, 128, device="cuda") * 3
ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
s1, i1 = ops.ip_topk_prefiltered(d, q_raw, k, 1.0, mode=ops.MODE_IP_NORMQ, workspace=ws)
print("fell back:", ops.prefilter_fell_back(ws))
lib = _lib.load()
a_s = np.zeros((nq, 64), np.float32); a_i = np.zeros((nq, 64), np.int64); kp = ctypes.c_int(0)
assert lib.ms_debug_prefilter_lists(ws.data_ptr(), n, nq, k, a_s.ctypes.data, a_i.ctypes.data, ctypes.byref(kp)) == 0
kp = kp.value
a_s = a_s.reshape(-1)[:nq * kp].reshape(nq, kp); a_i = a_i.reshape(-1)[:nq * kp].reshape(nq, kp)
qn = (q_raw / q_raw.norm(dim=1, keepdim=True))
ex = (d[torch.from_numpy(a_i.clip(0)).cuda().reshape(-1)].reshape(nq, kp, 128).double() * qn.double()[:, None, :]).sum(2).cpu().numpy()
print("kp", kp, "max |approx - exact| over candidates:", np.abs(ex - a_s)[a_i >= 0].max())
kth = np.sort(ex, axis=1)[:, ::-1][:, k - 1]
viol = ~(kth > a_s[:, kp - 1] + 2.5e-4 * 1.001)
print("queries failing the proof:", viol.sum(), "of", nq, "; lists not full:", (a_i[:, kp - 1] < 0).sum())
j = int(np.argmax(viol)) if viol.any() else 0
np.set_printoptions(linewidth=250, precision=5)
print("example query", j, "approx", a_s[j][:12], "...", a_s[j][-3:], "rows", a_i[j][:12], "exact", ex[j][:12], "kth", kth[j])
bad = np.abs(ex - a_s) > 1e-3
print("garbage entries:", bad.sum(), "in queries", np.unique(np.nonzero(bad)[0])[:20], "rows mod 32:", np.unique(a_i[bad] % 32)[:40], "tiles:", np.unique((a_i[bad] % 7840) // 32)[:20])
