"""Diagnostic (-DMS_STAMP builds): phase timeline of the few-query scan launch (ms_scan_body) and of its sample pass.
usage: MS_LIB_OVERRIDE=build/stamp/libmerizo_search_amd.so python3 tools/stamp_body.py ROWS,NQ [...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn

lib = _lib.load()
lib.ms_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
k = 10
names = ["entry", "queries ready", "tile 0 landed", "stream done", "lists written", "exit"]
for a in sys.argv[1:]:
    n, nq = (int(x) for x in a.split(","))
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    q_raw = torch.randn(nq, 128, device="cuda") * 3
    ws = ops.TopKWorkspace(d.device).get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    for _ in range(30): ops.ip_topk(d, q_raw, k, mode=ops.MODE_IP_NORMQ, workspace=ws, out=(out_s, out_i))
    torch.cuda.synchronize()
    words = 2 * 8 * 8 * 4096
    buf = np.zeros(words, dtype=np.uint64)
    assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
    for part, label in ((0, "scan"), (1, "sample pass")):
        full = buf[part * 8 * 8 * 4096:(part + 1) * 8 * 8 * 4096].reshape(-1, 8).astype(np.float64)
        st = full[:, :6]
        lastwg = full[(full[:, 6] > 0) & (full[:, 5] > 0)]
        st = st[(st[:, 0] > 0) & (st[:, 5] > 0)]
        if not len(st): continue
        t0 = st[:, 0].min()
        print(f"n={n} nq={nq} {label}: {len(st)} waves, span {(st[:, 5].max() - t0) / 100:.1f} us")
        for j, nm in enumerate(names):
            c = st[:, j][st[:, j] > 0]
            if len(c): print(f"    {nm:>14}: min {(c.min()-t0)/100:6.1f}  median {(np.median(c)-t0)/100:6.1f}  max {(c.max()-t0)/100:6.1f} us")
        rd = full[(full[:, 0] > 0) & (full[:, 5] > 0)][0::4]      # wave 0 of every workgroup
        if part == 0 and len(rd):
            c = rd[:, 1] - t0
            print(f"    wave 0 'queries ready / bound first seen': min {c.min()/100:.1f} median {np.median(c)/100:.1f} max {c.max()/100:.1f} us")
        if len(lastwg) and part == 0:
            t_l = lastwg[:, 4].max()
            print(f"    last workgroup: lists written {(t_l-t0)/100:.1f}, ticket taken + acquired {(lastwg[:,6].max()-t0)/100:.1f}, staged {(lastwg[:,7].max()-t0)/100:.1f}, exit {(lastwg[:,5].max()-t0)/100:.1f} us")
    del d
