"""Tensor-level front end of the C ABI: torch tensors in, torch tensors out, HIP underneath.

Every function runs on the tensors' device via libmerizo_search_amd.so on torch's current
stream and raises ``MerizoHipError`` when the library or a GPU is missing (no fallback).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import os

import numpy as np

from . import _lib
from ._lib import DIM, MODE_COSINE_RAW, MODE_COSINE_UNIT, MODE_IP_NORMQ, MODE_IP_PRENORM, MerizoHipError, check, ptr


class _on:
    """Device scope of one C-ABI call: the library launches on HIP's CURRENT device, so every entry
    point makes the tensors' device current for the call and hands over THAT device's current
    stream.  All tensor arguments must live on one device."""

    def __init__(self, *tensors):
        torch = _lib.require_gpu()
        devs = {t.device for t in tensors if t is not None}
        if len(devs) != 1:
            raise MerizoHipError(f"tensors of one call must share a device, got {sorted(str(d) for d in devs)}")
        self.device = devs.pop()
        if self.device.type != "cuda":
            raise MerizoHipError("expected CUDA/HIP tensors")
        self._guard = torch.cuda.device(self.device)
        self.stream = None

    def __enter__(self):
        import torch
        self._guard.__enter__()
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        return self

    def __exit__(self, *exc):
        return self._guard.__exit__(*exc)


def _f32_cuda(t, name: str, cols: Optional[int] = None):
    torch = _lib.require_gpu()
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MerizoHipError(f"{name}: expected a CUDA/HIP tensor")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise MerizoHipError(f"{name}: expected a contiguous float32 tensor")
    if cols is not None and (t.dim() != 2 or t.shape[1] != cols):
        raise MerizoHipError(f"{name}: expected shape [n,{cols}], got {tuple(t.shape)}")
    return t


def l2_normalize_rows_(x, eps: float = 1e-12):
    """In-place F.normalize(x) (reference dbsearch.py:303-304)."""
    _f32_cuda(x, "x", DIM)
    with _on(x) as dev:
        check(_lib.load().ms_l2_normalize_rows(ptr(x), x.shape[0], DIM, eps, dev.stream), "ms_l2_normalize_rows")
    return x


def l2_normalize_rows(x, eps: float = 1e-12, out=None):
    """F.normalize(x) out of place (reference dbsearch.py:303-304) -> a new tensor, or `out`."""
    torch = _lib.require_gpu()
    _f32_cuda(x, "x", DIM)
    y = torch.empty_like(x) if out is None else _f32_cuda(out, "out", DIM)
    if y.shape != x.shape or y.data_ptr() == x.data_ptr():
        raise MerizoHipError("l2_normalize_rows: out must have x's shape and not alias it")
    with _on(x, y) as dev:
        check(_lib.load().ms_l2_normalize_rows_to(ptr(x), ptr(y), x.shape[0], DIM, eps, dev.stream), "ms_l2_normalize_rows_to")
    return y


def row_inv_norms(x, eps: float = 1e-8):
    """1 / max(||row||, eps) for every database row (the DB half of cosine_similarity)."""
    torch = _lib.require_gpu()
    _f32_cuda(x, "x", DIM)
    inv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    with _on(x) as dev:
        check(_lib.load().ms_row_inv_norms(ptr(x), x.shape[0], DIM, eps, ptr(inv), dev.stream), "ms_row_inv_norms")
    return inv


class TopKWorkspace:
    """Scratch buffer for ms_ip_topk, grown on demand and reused across calls."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, n: int, nq: int, k: int):
        torch = _lib.require_gpu()
        with torch.cuda.device(self.device):          # the plan depends on the device's CU count
            need = int(_lib.load().ms_ip_topk_workspace_bytes(n, nq, k))
        if need == 0:
            raise MerizoHipError(f"ms_ip_topk_workspace_bytes rejected n={n} nq={nq} k={k}")
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.buf


def ip_topk(db, q, k: int, mode: int = MODE_IP_PRENORM, inv_norm=None, lengths=None, qlen=None,
            mincov: float = 0.0, row_offset: int = 0, workspace=None, out=None):
    """Exact top-k of q [nq,128] against db [n,128] -> (scores f32 [nq,k], idx i64 [nq,k]).
    workspace: a TopKWorkspace (grown on demand) or a uint8 tensor of ms_ip_topk_workspace_bytes;
    out: optional preallocated (scores, idx) tensors."""
    torch = _lib.require_gpu()
    _f32_cuda(db, "db", DIM)
    _f32_cuda(q, "q", DIM)
    n, nq = db.shape[0], q.shape[0]
    for name, t, size in (("inv_norm", inv_norm, n), ("lengths", lengths, n), ("qlen", qlen, nq)):
        if t is not None:
            _f32_cuda(t, name)
            if t.numel() != size:
                raise MerizoHipError(f"{name}: expected {size} elements, got {t.numel()}")
    ws = workspace if isinstance(workspace, torch.Tensor) else (workspace or TopKWorkspace(db.device)).get(n, nq, k)
    if out is None:
        out_s = torch.empty((nq, k), dtype=torch.float32, device=db.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=db.device)
    else:
        out_s, out_i = out
        if (tuple(out_s.shape) != (nq, k) or tuple(out_i.shape) != (nq, k) or out_s.dtype != torch.float32
                or out_i.dtype != torch.int64 or not out_s.is_contiguous() or not out_i.is_contiguous()):
            raise MerizoHipError("ip_topk: out must be contiguous (float32 [nq,k], int64 [nq,k]) tensors")
    with _on(db, q, inv_norm, lengths, qlen, ws, out_s, out_i) as dev:
        check(_lib.load().ms_ip_topk(ptr(db), n, row_offset, ptr(q), nq, k, mode, ptr(inv_norm), ptr(lengths), ptr(qlen),
                                     mincov, ptr(out_s), ptr(out_i), ptr(ws), ws.numel(), dev.stream), "ms_ip_topk")
    return out_s, out_i


def ip_topk_prepare(db, q, k: int, ws, mode: int = MODE_IP_PRENORM, inv_norm=None, lengths=None, qlen=None,
                    mincov: float = 0.0):
    """Stage 1 of ip_topk (k <= 64): query preparation + sample pass."""
    with _on(db, q, inv_norm, lengths, qlen, ws) as dev:
        check(_lib.load().ms_ip_topk_prepare(ptr(db), db.shape[0], ptr(q), q.shape[0], k, mode, ptr(inv_norm), ptr(lengths),
                                             ptr(qlen), mincov, ptr(ws), ws.numel(), dev.stream), "ms_ip_topk_prepare")


def ip_topk_scan(db, q, k: int, ws, mode: int = MODE_IP_PRENORM, inv_norm=None, lengths=None, qlen=None,
                 mincov: float = 0.0):
    """Stage 2 of ip_topk (k <= 64): the one launch of the fused score + top-k scan kernel (bench timing)."""
    with _on(db, q, inv_norm, lengths, qlen, ws) as dev:
        check(_lib.load().ms_ip_topk_scan(ptr(db), db.shape[0], ptr(q), q.shape[0], k, mode, ptr(inv_norm), ptr(lengths),
                                          ptr(qlen), mincov, ptr(ws), ws.numel(), dev.stream), "ms_ip_topk_scan")


def ip_topk_finish(n: int, nq: int, k: int, ws, out_s, out_i, row_offset: int = 0):
    """Stage 3 of ip_topk (k <= 64): merge the per-stream lists into the outputs."""
    with _on(ws, out_s, out_i) as dev:
        check(_lib.load().ms_ip_topk_finish(n, row_offset, nq, k, ptr(out_s), ptr(out_i), ptr(ws), ws.numel(),
                                            dev.stream), "ms_ip_topk_finish")


class PrefilterWorkspace(TopKWorkspace):
    """Scratch buffer for ms_ip_topk_prefiltered (the plain search's workspace plus the candidate lists)."""

    def get(self, n: int, nq: int, k: int):
        torch = _lib.require_gpu()
        with torch.cuda.device(self.device):
            need = int(_lib.load().ms_ip_topk_prefiltered_workspace_bytes(n, nq, k))
        if need == 0:
            raise MerizoHipError(f"ms_ip_topk_prefiltered_workspace_bytes rejected n={n} nq={nq} k={k}")
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.buf


PF_BF16X3, PF_F16X2, PF_F16X1 = _lib.PF_BF16X3, _lib.PF_F16X2, _lib.PF_F16X1
_PF_NAMES = {"bf16x3": PF_BF16X3, "f16x2": PF_F16X2, "f16x1": PF_F16X1}


def pf_default_format() -> int:
    """The image the driver builds: fp16 rows (256 B per row; MS_PF_F16X2 / MS_PF_F16X1 share it -- which of the two arithmetics runs
    is decided per database by pf_choose_format unless MS_PF_FORMAT names one); MS_PF_FORMAT=bf16x3|f16x2|f16x1|auto overrides
    (A/B runs; MS_PF_BF16X3 is also what the engine falls back to for databases whose rows leave the fp16 range)."""
    name = os.environ.get("MS_PF_FORMAT", "auto").lower()
    if name == "auto":
        return PF_F16X2
    if name not in _PF_NAMES:
        raise MerizoHipError(f"MS_PF_FORMAT={name}: expected one of {sorted(_PF_NAMES) + ['auto']}")
    return _PF_NAMES[name]


def pf_format_is_auto() -> bool:
    return os.environ.get("MS_PF_FORMAT", "auto").lower() == "auto"


class PfImage:
    """An MFMA-ready image of a resident database (ms_pf_build_image) with the arithmetic it was built for.  `data`: uint8 tensor;
    `format`: PF_BF16X3 (512 B per row) or PF_F16X2 / PF_F16X1 (ONE fp16 image, 256 B per row: `as_format` switches between the two)."""
    __slots__ = ("data", "format", "n")

    def __init__(self, data, fmt: int, n: int):
        self.data, self.format, self.n = data, int(fmt), int(n)

    def numel(self) -> int:
        return self.data.numel()

    def as_format(self, fmt: int) -> "PfImage":
        if fmt == self.format:
            return self
        if PF_BF16X3 in (fmt, self.format):
            raise MerizoHipError("the split-bf16 image and the fp16 image are different buffers: build the one you want")
        return PfImage(self.data, fmt, self.n)


def pf_err_coef(fmt: int) -> float:
    """E of the format: |approximate - exact| <= E |row| |q| (ms_pf_err_coef)."""
    return float(_lib.load().ms_pf_err_coef(int(fmt)))


def pf_build_image(db, fmt=None, row_norm_bound=None, out=None) -> PfImage:
    """The image of a resident database for the prefiltered search (ms_pf_build_image), built once next to the fp32 rows:
    fp16 rows (PF_F16X2 / PF_F16X1, 256 B per row; needs the rows' norm bound -- measured here when not given) or split-bf16
    (PF_BF16X3, 512 B per row)."""
    torch = _lib.require_gpu()
    _f32_cuda(db, "db", DIM)
    fmt = pf_default_format() if fmt is None else int(fmt)
    n = db.shape[0]
    need = int(_lib.load().ms_pf_image_bytes(n, fmt))
    if fmt != PF_BF16X3 and row_norm_bound is None:
        row_norm_bound = (float(1.0 / row_inv_norms(db, 1e-30).min()) * (1.0 + 1e-6)) if n > 0 else 1.0
    img = torch.empty(max(need, 16), dtype=torch.uint8, device=db.device) if out is None else out
    if img.dtype != torch.uint8 or img.numel() < need or not img.is_contiguous():
        raise MerizoHipError("pf_build_image: out must be a contiguous uint8 tensor of ms_pf_image_bytes(n, fmt)")
    with _on(db, img) as dev:
        check(_lib.load().ms_pf_build_image(ptr(db), n, fmt, float(row_norm_bound or 0.0), ptr(img), dev.stream), "ms_pf_build_image")
    return PfImage(img, fmt, n)


def pf_choose_format(db, image: PfImage, row_norm_bound: float, nsample: int = 256, k: int = 10, max_flagged: float = 0.02) -> PfImage:
    """F16X1 or F16X2 for a resident database, decided ONCE by running the search itself: `nsample` rows of the database, evenly
    spaced, are searched as queries (top-k) with the one-instruction arithmetic (MS_PF_F16X1: E = 1.05e-3 |row||q|); if more than
    `max_flagged` of them fail their proof -- the database has families of rows within ~1e-3 of each other around its own entries,
    as real embedding databases do -- the two-instruction arithmetic (MS_PF_F16X2, E = 5.5e-4) is used from then on (the SAME
    image: nothing is rebuilt).  The answers are exact either way; this only picks the faster of two exact paths.  One sync."""
    torch = _lib.require_gpu()
    if image is None or image.format == PF_BF16X3:
        return image
    n = db.shape[0]
    if not prefilter_serves(n, max(nsample, 65), k):
        return image
    idx = (torch.arange(nsample, dtype=torch.int64, device=db.device) * (n - 1)) // max(1, nsample - 1)      # (integers: float32 cannot count 45 M rows)
    q = db[idx].contiguous()
    ws = PrefilterWorkspace(db.device).get(n, nsample, k)
    ip_topk_prefiltered(db, q, k, row_norm_bound, mode=MODE_IP_NORMQ, workspace=ws, image=image.as_format(PF_F16X1))
    flagged = prefilter_flagged(ws)
    return image.as_format(PF_F16X1 if flagged <= max_flagged * nsample else PF_F16X2)


def small_batch_thresholds():
    """(fused_merge_max_nq, inkernel_norm_max_nq) as the loaded library applies them (ms_small_batch_thresholds)."""
    import ctypes
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    _lib.load().ms_small_batch_thresholds(ctypes.byref(a), ctypes.byref(b))
    return int(a.value), int(b.value)


def prefilter_serves(n: int, nq: int, k: int, image=None) -> bool:
    """The shapes ms_ip_topk_prefiltered serves itself (everything else it hands to ms_ip_topk): ONE statement of the rule for the
    engine, the bench and the tests (the library applies the same in pf_layout).  More than 64 queries, k <= 48, >= 65,536 rows; and,
    over an fp16 `image` (a PfImage, or a format), ANY number of queries from ms_pf_few_min_rows(nq) rows on (1M rows for 1..32 queries:
    the HBM-bound regime at half the bytes; 200k rows for 33..64: two query tiles of fp32 matrix work against one fp16 pass)."""
    if k > _lib.PREFILTER_MAX_K or n < _lib.PREFILTER_MIN_ROWS:
        return False
    if nq > 64:
        return True
    fmt = image.format if isinstance(image, PfImage) else image
    return fmt in (PF_F16X2, PF_F16X1) and n >= int(_lib.load().ms_pf_few_min_rows(int(nq)))


def _pf_args(db, image, q, mode, lengths, qlen):
    _f32_cuda(db, "db", DIM)
    _f32_cuda(q, "q", DIM)
    if mode not in (MODE_IP_PRENORM, MODE_IP_NORMQ, MODE_COSINE_UNIT):
        raise MerizoHipError("ip_topk_prefiltered: MODE_IP_PRENORM, MODE_IP_NORMQ or MODE_COSINE_UNIT")
    if mode != MODE_COSINE_UNIT and (lengths is not None or qlen is not None):
        raise MerizoHipError("ip_topk_prefiltered: lengths / qlen go with MODE_COSINE_UNIT")
    if image is not None and (not isinstance(image, PfImage) or image.n != db.shape[0] or
                              image.data.numel() < int(_lib.load().ms_pf_image_bytes(db.shape[0], image.format))):
        raise MerizoHipError("ip_topk_prefiltered: image is not pf_build_image(db)")
    for name, t, size in (("lengths", lengths, db.shape[0]), ("qlen", qlen, q.shape[0])):
        if t is not None:
            _f32_cuda(t, name)
            if t.numel() != size:
                raise MerizoHipError(f"{name}: expected {size} elements, got {t.numel()}")


def ip_topk_prefiltered(db, q, k: int, row_norm_bound: float = 1.0, mode: int = MODE_IP_PRENORM, row_offset: int = 0,
                        workspace=None, out=None, image=None, lengths=None, qlen=None, mincov: float = 0.0):
    """ip_topk with the same results bit for bit, several times faster for more than 64 queries and k <= 48: 16-bit prefilter
    scan (over `image` = pf_build_image(db) -- fp16 or split-bf16 -- when given, else splitting the rows in registers), exact re-scoring, per-query proof,
    and an exact fp32 pass over the queries whose proof failed (include/merizo_search_amd.h).  row_norm_bound: an upper bound on
    every row's L2 norm (1.0 + 1e-6 for unit rows).  workspace: a PrefilterWorkspace or a uint8 tensor of
    ms_ip_topk_prefiltered_workspace_bytes."""
    torch = _lib.require_gpu()
    _pf_args(db, image, q, mode, lengths, qlen)
    n, nq = db.shape[0], q.shape[0]
    ws = workspace if isinstance(workspace, torch.Tensor) else (workspace or PrefilterWorkspace(db.device)).get(n, nq, k)
    if out is None:
        out_s = torch.empty((nq, k), dtype=torch.float32, device=db.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=db.device)
    else:
        out_s, out_i = out
    idata, ifmt = (image.data, image.format) if image is not None else (None, 0)
    with _on(db, q, ws, out_s, out_i, idata, lengths, qlen) as dev:
        check(_lib.load().ms_ip_topk_prefiltered(ptr(db), ptr(idata), ifmt, n, row_offset, ptr(q), nq, k, mode, ptr(lengths), ptr(qlen), mincov,
                                                 float(row_norm_bound), ptr(out_s), ptr(out_i), ptr(ws), ws.numel(), dev.stream),
              "ms_ip_topk_prefiltered")
    return out_s, out_i


def ip_topk_prefiltered_stage(stage: str, db, q, k: int, ws, row_norm_bound: float = 1.0, mode: int = MODE_IP_PRENORM,
                              out=None, row_offset: int = 0, image=None, lengths=None, qlen=None, mincov: float = 0.0):
    """One stage of ip_topk_prefiltered ('prepare', 'scan', 'finish'): lets a bench time the scan launch alone."""
    lib = _lib.load()
    n, nq = db.shape[0], q.shape[0]
    idata, ifmt = (image.data, image.format) if image is not None else (None, 0)
    with _on(db, q, ws, idata, lengths, qlen) as dev:
        if stage == "prepare":
            check(lib.ms_ip_topk_prefiltered_prepare(ptr(db), ptr(idata), ifmt, n, ptr(q), nq, k, mode, ptr(lengths), ptr(qlen), mincov,
                                                     float(row_norm_bound), ptr(ws), ws.numel(), dev.stream), "ms_ip_topk_prefiltered_prepare")
        elif stage == "scan":
            check(lib.ms_ip_topk_prefiltered_scan(ptr(db), ptr(idata), ifmt, n, ptr(q), nq, k, mode, ptr(lengths), ptr(qlen), mincov,
                                                  float(row_norm_bound), ptr(ws), ws.numel(), dev.stream), "ms_ip_topk_prefiltered_scan")
        else:
            out_s, out_i = out
            check(lib.ms_ip_topk_prefiltered_finish(ptr(db), ptr(idata), ifmt, n, row_offset, ptr(q), nq, k, mode, ptr(lengths), ptr(qlen), mincov,
                                                    float(row_norm_bound), ptr(out_s), ptr(out_i), ptr(ws), ws.numel(), dev.stream),
                  "ms_ip_topk_prefiltered_finish")


def prefilter_flagged(ws) -> int:
    """Diagnostics (synchronises): how many queries of the last prefiltered search on this workspace needed the exact pass."""
    import ctypes
    g, e, c = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_uint(0)
    check(_lib.load().ms_debug_prefilter_state(ptr(ws), ctypes.byref(g), ctypes.byref(e), ctypes.byref(c)), "ms_debug_prefilter_state")
    return int(c.value) if e.value != 0 else 0


def prefilter_poison(ws, slot_counter: int, ticket: int) -> None:
    """Diagnostics for the tests (synchronises): overwrite the compaction's slot counter and ticket in the library-owned block of this
    workspace -- the state an aborted launch or a second stream on the workspace would leave (ms_debug_prefilter_poison).  The next
    prefiltered search on `ws` must fail loudly."""
    check(_lib.load().ms_debug_prefilter_poison(ptr(ws), int(slot_counter) & 0xFFFFFFFF, int(ticket) & 0xFFFFFFFF), "ms_debug_prefilter_poison")


def device_pci_bus_id(device=None) -> str:
    """PCI bus id of `device` (default: torch's current device), e.g. '0000:c1:00.0' (ms_device_pci_bus_id)."""
    import ctypes
    torch = _lib.require_gpu()
    buf = ctypes.create_string_buffer(64)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(_lib.load().ms_device_pci_bus_id(buf, 64), "ms_device_pci_bus_id")
    return buf.value.decode()


def prefilter_fell_back(ws) -> bool:
    """Diagnostics (synchronises): did any query of the last prefiltered search on this workspace need the exact pass?"""
    return prefilter_flagged(ws) > 0


def topk_merge(scores, idx):
    """Merge [S,nq,k] sorted result lists (shards or blocks) into [nq,k]."""
    torch = _lib.require_gpu()
    if scores.dim() != 3 or idx.shape != scores.shape:
        raise MerizoHipError("topk_merge: expected scores/idx of shape [S,nq,k]")
    scores = scores.contiguous()
    idx = idx.contiguous()
    if scores.dtype != torch.float32 or idx.dtype != torch.int64 or not scores.is_cuda:
        raise MerizoHipError("topk_merge: expected float32 / int64 CUDA tensors")
    S, nq, k = scores.shape
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    with _on(scores, idx) as dev:
        check(_lib.load().ms_topk_merge(ptr(scores), ptr(idx), S, nq, k, ptr(out_s), ptr(out_i), dev.stream),
              "ms_topk_merge")
    return out_s, out_i


def topk_merge_packed(gathered, S: int, nq: int, k: int, idx_offset: int, out_s, out_i):
    """Merge S packed result blocks [scores f32 nq*k | pad | rows i64 nq*k] laid out back to back in the
    uint8 buffer ``gathered`` (the all-gather output) without unpacking them."""
    torch = _lib.require_gpu()
    if gathered.dtype != torch.uint8 or not gathered.is_cuda or not gathered.is_contiguous():
        raise MerizoHipError("topk_merge_packed: expected a contiguous uint8 CUDA buffer")
    stride = gathered.numel() // S
    if stride * S != gathered.numel() or stride < idx_offset + 8 * nq * k:
        raise MerizoHipError("topk_merge_packed: buffer size does not match S blocks of nq*k results")
    base = gathered.data_ptr()
    with _on(gathered, out_s, out_i) as dev:
        check(_lib.load().ms_topk_merge_strided(base, base + idx_offset, stride, stride, S, nq, k, ptr(out_s), ptr(out_i),
                                                dev.stream), "ms_topk_merge_strided")
    return out_s, out_i


class EgnnEncoder:
    """The Foldclass structure encoder on one GPU: prepared weights + positional table.

    ``embed(list of [N,3] arrays)`` -> float32 [B,128] tensor on the device.  Replaces
    ``network(x)`` of the reference (dbsearch.py:97-98, :299-301; makedb.py:75-79) for a whole
    ragged batch per launch.
    """

    def __init__(self, weights: np.ndarray, pe: np.ndarray, device="cuda:0"):
        torch = _lib.require_gpu()
        lib = _lib.load()
        self.device = torch.device(device)
        nfl = int(lib.ms_egnn_weight_floats())
        weights = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
        if weights.size != nfl:
            raise MerizoHipError(f"encoder weights: expected {nfl} floats, got {weights.size}")
        pe = np.ascontiguousarray(pe, dtype=np.float32).reshape(-1, DIM)
        with torch.cuda.device(self.device):
            w_dev = torch.from_numpy(weights).to(self.device)
            self.pe = torch.from_numpy(pe).to(self.device)
            self.prepared = torch.empty(int(lib.ms_egnn_prepared_bytes()), dtype=torch.uint8, device=self.device)
            check(lib.ms_egnn_prepare_weights(ptr(w_dev), ptr(self.prepared),
                                              torch.cuda.current_stream(self.device).cuda_stream), "ms_egnn_prepare_weights")
            torch.cuda.current_stream(self.device).synchronize()
        self.max_len = pe.shape[0]
        self._ws = None
        self._stage = self._stage_dev = self._stage_free = None

    def embed(self, coords_list: Sequence[np.ndarray]):
        torch = _lib.require_gpu()
        lib = _lib.load()
        nb = len(coords_list)
        if nb == 0:
            return torch.empty((0, DIM), dtype=torch.float32, device=self.device)
        lens = np.array([int(np.asarray(c).shape[0]) for c in coords_list], dtype=np.int64)
        if lens.min() < 1:
            raise MerizoHipError("embed: empty structure")
        if lens.max() > self.max_len:
            raise MerizoHipError(f"embed: structure of {int(lens.max())} residues exceeds the positional table "
                                 f"({self.max_len}); the reference fails here too (nndef_fold_egnn_embed.py:12)")
        total = int(lens.sum())
        # offsets and coordinates go to the device in ONE asynchronous copy out of a pinned staging buffer (two pageable
        # copies cost a query of a few domains ~50 us of host time): [int32 offsets (nb + 1) | pad | float32 coords (total x 3)]
        head = ((nb + 1) * 4 + 15) // 16 * 16
        nbytes = head + total * 12
        with torch.cuda.device(self.device):
            if self._stage is None or self._stage.numel() < nbytes:
                cap = max(nbytes, 1 << 16) * 2
                self._stage = torch.empty(cap, dtype=torch.uint8).pin_memory()
                self._stage_dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
                self._stage_free = None
            if self._stage_free is not None:
                self._stage_free.synchronize()            # the previous call's copy has left the staging buffer
            host = self._stage.numpy()
            offsets = host[: (nb + 1) * 4].view(np.int32)
            offsets[0] = 0
            offsets[1:] = np.cumsum(lens)
            flat = host[head: nbytes].view(np.float32).reshape(total, 3)
            pos = 0
            for c, n in zip(coords_list, lens):
                flat[pos: pos + int(n)] = np.asarray(c, dtype=np.float32).reshape(-1, 3)
                pos += int(n)
            stream = torch.cuda.current_stream(self.device)
            self._stage_dev[:nbytes].copy_(self._stage[:nbytes], non_blocking=True)
            self._stage_free = stream.record_event()
            offs_ptr = self._stage_dev.data_ptr()
            coords_ptr = offs_ptr + head
            need = int(lib.ms_egnn_workspace_bytes(nb, total, int((lens * lens).sum())))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            out = torch.empty((nb, DIM), dtype=torch.float32, device=self.device)
            # (`offsets` in the pinned buffer is only read by the host during the call itself)
            check(lib.ms_egnn_embed(ptr(self.prepared), ptr(self.pe), self.max_len, coords_ptr, offs_ptr,
                                    offsets.ctypes.data, nb, ptr(out), ptr(self._ws), self._ws.numel(), stream.cuda_stream),
                  "ms_egnn_embed")
        return out
