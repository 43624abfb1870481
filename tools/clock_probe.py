"""Diagnostic (stamp build: MS_LIB_OVERRIDE=.../build/stamp/libmerizo_search_amd.so): in-kernel clock of the scan launch inside
different step loops -- does what runs around the scan (events, small kernels, synchronisation) change the clock the chip holds?"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merizo_search_amd import ops, _lib
from merizo_search_amd.foldclass import synthetic as syn

lib = _lib.load()
lib.ms_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n, nq, k = 1_000_000, 256, 10
d = syn.device_database(n, 0, 0, "cuda:0", normalize=True)
q_raw = torch.randn(nq, 128, device="cuda") * 3
q = torch.empty_like(q_raw)
ws = ops.TopKWorkspace(d.device).get(n, nq, k)
out_s = torch.empty(nq, k, device="cuda"); out_i = torch.empty(nq, k, dtype=torch.int64, device="cuda")


def clock():
    words = 8 * 8 * 4096
    buf = np.zeros(words, dtype=np.uint64)
    assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
    st = buf.reshape(-1, 8, 8)[:, :4, :].reshape(-1, 8); st = st[st[:, 2] > 0]
    cyc, rt, nt = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64), st[:, 2].astype(np.float64)
    return np.median(cyc / rt) * 0.1, np.median(cyc / nt), np.max(rt) / 100


def loop(name, steps, normalize=True, events=False, sync_every=0, sleep_us=0):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(steps):
        if normalize: ops.l2_normalize_rows(q_raw, 1e-12, out=q)
        ops.ip_topk_prepare(d, q, k, ws)
        if events: evs[s][0].record()
        ops.ip_topk_scan(d, q, k, ws)
        if events: evs[s][1].record()
        ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)
        if sync_every and (s + 1) % sync_every == 0: torch.cuda.synchronize()
        if sleep_us: time.sleep(sleep_us * 1e-6)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    ghz, cpt, wmax = clock()
    ev = " scan (events) %.1f us" % (np.mean([a.elapsed_time(b) for a, b in evs[steps // 2:]]) * 1e3) if events else ""
    print(f"{name:52s} {el / steps * 1e6:7.1f} us/step | last scan launch: clock {ghz:.3f} GHz, {cpt:.0f} cycles/tile, slowest wave {wmax:.1f} us{ev}", flush=True)


q.copy_(q_raw / q_raw.norm(dim=1, keepdim=True))
loop("first 30 steps", 30)
loop("200 steps", 200)
loop("200 steps + events around the scan", 200, events=True)
loop("200 steps, no normalise kernel", 200, normalize=False)
loop("2000 steps", 2000)
loop("200 steps, host sync every step", 200, sync_every=1)
loop("200 steps + events (again)", 200, events=True)

# per-XCD spread of the last launch: workgroups b and b + 8 share an XCD (round-robin dispatch)
words = 8 * 8 * 4096
buf = np.zeros(words, dtype=np.uint64)
assert lib.ms_debug_stamps(buf.ctypes.data, words) == 0
allw = buf.reshape(-1, 8, 8)
act = np.nonzero(allw[:, 0, 2] > 0)[0]
print("per XCD label (workgroup id % 8): wave time median / max (us), clock (GHz), cycles per tile")
for x in range(8):
    sel = act[act % 8 == x]
    st = allw[sel, :4, :].reshape(-1, 8)
    cyc, rt, nt = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64), st[:, 2].astype(np.float64)
    print(f"  xcd {x}: {np.median(rt) / 100:7.1f} / {np.max(rt) / 100:7.1f} | {np.median(cyc / rt) * 0.1:.3f} | {np.median(cyc / nt):.0f} (min {np.min(cyc / nt):.0f} max {np.max(cyc / nt):.0f})")
st = allw[act, :4, :].reshape(-1, 8)
rt = st[:, 1].astype(np.float64) / 100
print("all waves: wave time percentiles 1/25/50/75/99/100: " + " ".join("%.1f" % v for v in np.percentile(rt, [1, 25, 50, 75, 99, 100])))
w0 = allw[act, 0, 1].astype(np.float64) / 100; w3 = allw[act, 3, 1].astype(np.float64) / 100
print("wave 0 (shares its SIMD with the loader) median %.1f us, wave 3 median %.1f us" % (np.median(w0), np.median(w3)))
