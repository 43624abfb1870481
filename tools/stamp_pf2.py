"""Diagnostic (-DMS_STAMP build): where a compute wave of the split-image prefilter scan (ms_scan_pf2_kernel) spends its cycles.
usage: MS_LIB_OVERRIDE=build/stamp/libmerizo_search_amd.so python3 tools/stamp_pf2.py ROWS,NQ,K [...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn

lib = _lib.load()
lib.ms_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1_000_000, 256, 10)]
for n, nq, k in cases:
    d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
    qq = torch.randn(nq, 128, device="cuda"); qq = qq / qq.norm(dim=1, keepdim=True)
    img = ops.pf_build_image(d)
    ws = ops.PrefilterWorkspace(d.device).get(n, nq, k)
    out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")
    prep = lambda: ops.ip_topk_prefiltered_stage("prepare", d, qq, k, ws, image=img)
    scan = lambda: ops.ip_topk_prefiltered_stage("scan", d, qq, k, ws, image=img)
    fin = lambda: ops.ip_topk_prefiltered_stage("finish", d, qq, k, ws, out=(out_s, out_i), image=img)
    for _ in range(30):
        prep(); scan(); fin()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    prep(); e0.record(); scan(); e1.record(); torch.cuda.synchronize()
    words = 8 * 8 * 4096
    buf = np.zeros(words, dtype=np.uint64)
    assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
    st = buf.reshape(-1, 8, 8).reshape(-1, 8)
    st = st[st[:, 2] > 0]
    f = lambda x: x.astype(np.float64)
    cyc, rt, nt = f(st[:, 0]), f(st[:, 1]), f(st[:, 2])
    lg, flow = f(st[:, 3] >> np.uint64(32)), f(st[:, 3] & np.uint64(0xFFFFFFFF))
    vis, nvis, chain = f(st[:, 4]), f(st[:, 5]), f(st[:, 6])
    hist, sync = f(st[:, 7] >> np.uint64(32)), f(st[:, 7] & np.uint64(0xFFFFFFFF))
    print(f"n={n} nq={nq} k={k}: scan {e0.elapsed_time(e1)*1e3:.1f} us | waves {len(st)} tiles/wave {nt.mean():.0f} | cycles/tile median {np.median(cyc/nt):.0f} "
          f"max {np.max(cyc/nt):.0f} | clock {np.median(cyc/rt)*0.1:.3f} GHz | wave time max {rt.max()/100:.1f} us median {np.median(rt)/100:.1f} us")
    print(f"   per tile (median over waves): lgkm wait {np.median(lg/nt):.0f}, own pieces (in-chain vmcnt) {np.median(flow/nt):.0f}, issue+vmcnt+landed {np.median(sync/nt):.0f}, "
          f"chain(+frag reads) {np.median(chain/nt):.0f}, visits {np.median(vis/nt):.0f} ({100*nvis.sum()/nt.sum():.1f} % of tiles x qtiles... {vis.sum()/max(nvis.sum(),1):.0f} cycles each), hist {np.median(hist/nt):.0f}")
    allw = buf.reshape(-1, 8, 8)
    act = allw[:, 0, 2] > 0
    for w in range(8):
        ww = allw[act, w, :]
        ww = ww[ww[:, 2] > 0]
        if not len(ww): continue
        ntw = f(ww[:, 2])
        print(f"   wave {w}: arrival wait/tile {np.median(f(ww[:, 7] & np.uint64(0xFFFFFFFF)) / ntw):.0f}  chain/tile {np.median(f(ww[:, 6]) / ntw):.0f}  "
              f"visits/tile {np.median(f(ww[:, 4]) / ntw):.0f}  cycles/tile {np.median(f(ww[:, 0]) / ntw):.0f}")
    del d, img
