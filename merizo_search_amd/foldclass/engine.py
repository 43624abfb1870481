"""The compute engine behind the Foldclass drivers.

``HipEngine`` is the product: every numeric step runs in libmerizo_search_amd.so on an
MI355X.  There is NO CPU engine in this package -- asking for device "cpu" fails loudly
(use the reference implementation for CPU runs).  The driver code (dbsearch.py, makedb.py)
only talks to the small interface below, so the test-suite can drive the same host logic
with an oracle-backed engine that lives under tests/ (test infrastructure, never shipped).

Engine interface (tensors are torch tensors on the engine's device):
    embed(list of float32 [N,3] arrays)                     -> float32 [B,128]
    to_device(numpy array)                                  -> tensor
    normalize_(x, eps)                                      -> x, rows L2-normalised in place
    normalized(x, eps)                                      -> new tensor, rows L2-normalised (F.normalize)
    cosine_rows(db)                                         -> the resident form of a RAW `.pt` matrix for cosine_topk
    cosine_topk(rows, q, k, lengths, qlen, mincov, row_offset[, pf_image]) -> (scores [nq,k], idx int64 [nq,k])
    ip_topk(db, q, k, row_offset, normalize_queries[, row_norm_bound, pf_image]) -> (scores [nq,k], idx int64 [nq,k]);
                                                               normalize_queries: q raw, F.normalize (eps 1e-12) fused into the call
    pf_image(db) / lazy_pf_image(db)                        -> the fp16 image of a resident matrix for the prefiltered search (built on first use)
                                                               (None when HBM has no room for it), built once per database
    topk_merge(scores [S,nq,k], idx [S,nq,k])               -> (scores [nq,k], idx [nq,k])
    merge_gathered(PackedExchange)                          -> (scores [nq,k], idx [nq,k])  multi-rank merge
    upload_rows(matrix, lo, hi)                             -> rows [lo,hi) of a host matrix as ONE device tensor
    device_blocks(blocks)                                   -> iterator of device tensors (out-of-core streaming)
    resident_budget(nq, k)                                  -> bytes a resident matrix may occupy
"""
from __future__ import annotations

import logging
import os
import sys
from typing import Optional, Sequence

import numpy as np

from . import weights as W

logger = logging.getLogger(__name__)


def resolve_device(device) -> str:
    """Map the reference's -d values onto this build: 'cuda' / 'cuda:N' select the MI355X
    (PyTorch-ROCm calls the HIP device 'cuda'); 'cpu' and 'mps' are refused."""
    name = str(device)
    if name.startswith("cuda") or name.startswith("hip"):
        return name.replace("hip", "cuda") if ":" in name else "cuda:0"
    logger.error("device '%s' is not supported by merizo_search_amd: the embed-and-search path runs on an "
                 "MI355X only (pass -d cuda); use the reference implementation for CPU runs." % name)
    sys.exit(1)


class LazyPfImage:
    """The prefilter's image of a resident matrix, built on first use (HipEngine.lazy_pf_image)."""

    def __init__(self, engine, db, row_norm_bound=None):
        self._engine, self._db, self._bound, self._image, self._built = engine, db, row_norm_bound, None, False
        self._few = 0              # few-query searches seen so far on a database large enough for the image to pay

    @property
    def built(self) -> bool:
        return self._built

    def want(self, n: int, nq: int, k: int):
        """The image for a batch of nq queries, building it when that pays: at once for a batch the prefilter serves without an
        fp16 image (more than 64 queries), from the THIRD few-query search on a database large enough for the HBM-bound regime over
        the image (one pass over the rows to build it = about one and a half searches) -- a CLI run with one or two query domains
        never pays for it."""
        if self._built:
            return self._image
        ops = self._engine._ops
        if ops.prefilter_serves(n, nq, k):
            return self.get()
        if ops.prefilter_serves(n, nq, k, ops.pf_default_format()):
            self._few += 1
            if self._few >= 3:
                return self.get()
        return None

    def get(self):
        if not self._built:
            self._image = self._engine.pf_image(self._db, self._bound)
            self._built = True
        return self._image


class HipEngine:
    name = "hip"

    def __init__(self, device="cuda:0", state_dict: Optional[dict] = None):
        from .. import _lib, ops
        self._ops = ops
        self.torch = _lib.require_gpu()
        self.device = self.torch.device(resolve_device(device))
        self._encoder = None
        self._state_dict = state_dict
        self._ws = ops.TopKWorkspace(self.device)
        self._pws = ops.PrefilterWorkspace(self.device)

    # -- encoder ---------------------------------------------------------------------
    def load_weights(self, state_dict: dict) -> None:
        self._state_dict = state_dict
        self._encoder = None

    def _get_encoder(self):
        if self._encoder is None:
            if self._state_dict is None:
                raise RuntimeError("no encoder weights loaded (network_setup does this)")
            weights, pe = W.pack_state_dict(self._state_dict)
            self._encoder = self._ops.EgnnEncoder(weights, pe, self.device)
        return self._encoder

    def embed(self, coords_list: Sequence[np.ndarray], max_batch_sq: int = 64_000_000):
        """Embed structures in ragged launches of at most `max_batch_sq` residue pairs each."""
        enc = self._get_encoder()
        outs, batch, acc = [], [], 0
        for c in coords_list:
            n = int(np.asarray(c).shape[0])
            if batch and acc + n * n > max_batch_sq:
                outs.append(enc.embed(batch)); batch, acc = [], 0
            batch.append(c); acc += n * n
        if batch:
            outs.append(enc.embed(batch))
        return outs[0] if len(outs) == 1 else self.torch.cat(outs, 0)

    # -- search ----------------------------------------------------------------------
    def to_device(self, array):
        t = array if isinstance(array, self.torch.Tensor) else self.torch.from_numpy(np.ascontiguousarray(array))
        return t.to(self.device).contiguous()

    def normalize_(self, x, eps: float = 1e-12):
        return self._ops.l2_normalize_rows_(x, eps)

    def normalized(self, x, eps: float = 1e-12):
        return self._ops.l2_normalize_rows(x, eps)

    def row_inv_norms(self, db, eps: float = 1e-8):
        return self._ops.row_inv_norms(db, eps)

    def cosine_rows(self, db):
        """Resident form of a RAW `.pt` matrix for cosine_topk: its rows are L2-normalised IN PLACE once
        (x / max(|x|, 1e-8): the row half of F.cosine_similarity's normalise-both-then-dot, dbsearch.py:78), so that
        every search runs the scan in MS_MODE_COSINE_UNIT, at the inner-product rate.  The raw matrix stays on disk."""
        return self._ops.l2_normalize_rows_(db, 1e-8)

    UNIT_ROW_BOUND = 1.0 + 1e-5      # |row| of rows normalised in fp32 (cosine_rows; dbfname_IP holds such rows too)

    def cosine_topk(self, rows, q, k, lengths=None, qlen=None, mincov: float = 0.0, row_offset: int = 0, pf_image=None):
        """search_query_against_db's arithmetic on rows normalised once (cosine_rows).  pf_image (`pf_image(rows)`, built when the
        database became resident): batches of more than 64 queries take the prefiltered search, same results bit for bit."""
        ops = self._ops
        image = self._image_for(pf_image, rows.shape[0], q.shape[0], k)
        if image is not None:
            return ops.ip_topk_prefiltered(rows, q, k, self.UNIT_ROW_BOUND, mode=ops.MODE_COSINE_UNIT, row_offset=row_offset,
                                           workspace=self._pws, image=image, lengths=lengths, qlen=qlen, mincov=mincov)
        return ops.ip_topk(rows, q, k, mode=ops.MODE_COSINE_UNIT, lengths=lengths, qlen=qlen, mincov=mincov,
                           row_offset=row_offset, workspace=self._ws)

    def ip_topk(self, db, q, k, row_offset: int = 0, normalize_queries: bool = False, row_norm_bound=None, pf_image=None):
        """index.search of dbsearch.py:234-242.  row_norm_bound: an upper bound on the rows' L2 norms when the caller knows one
        (`row_norm_bound(db)` once per resident database): batches of more than 64 queries then take the prefiltered search
        (ms_ip_topk_prefiltered: same results bit for bit; the rows scanned with fp16 matrix instructions over `pf_image`, the fp16
        image of the resident database -- any number of queries from 1M rows on -- or split in registers without one).  Queries whose answer the
        prefilter cannot prove get an exact pass of their own inside the same call: no feedback loop, no switch."""
        ops = self._ops
        mode = ops.MODE_IP_NORMQ if normalize_queries else ops.MODE_IP_PRENORM
        image = self._image_for(pf_image, db.shape[0], q.shape[0], k) if row_norm_bound is not None else None
        if row_norm_bound is not None and (image is not None or ops.prefilter_serves(db.shape[0], q.shape[0], k)):
            return ops.ip_topk_prefiltered(db, q, k, float(row_norm_bound), mode=mode, row_offset=row_offset, workspace=self._pws,
                                           image=image)
        return ops.ip_topk(db, q, k, mode=mode, row_offset=row_offset, workspace=self._ws)

    def _image_for(self, pf_image, n: int, nq: int, k: int):
        """The image this batch is searched over, or None (fp32 scan / rows split in registers).  Large batches: the database's
        image in the arithmetic chosen for it.  <= 64 queries over an fp16 image of >= ms_pf_few_min_rows(nq) rows: the HBM-bound regime
        at half the bytes (33..64 queries: one fp16 pass instead of two query tiles of fp32 matrix work) -- up to 32 queries with the tighter two-instruction arithmetic (the matrix pipe has time to spare there)."""
        ops = self._ops
        if pf_image is None:
            return None
        if isinstance(pf_image, LazyPfImage):
            pf_image = pf_image.want(n, nq, k)
            if pf_image is None:
                return None
        if not ops.prefilter_serves(n, nq, k, pf_image):
            return None
        if nq <= 32 and pf_image.format == ops.PF_F16X1 and ops.pf_format_is_auto():      # (a forced MS_PF_FORMAT is what runs: A/B runs measure what they name)
            return pf_image.as_format(ops.PF_F16X2)
        return pf_image

    def pf_image(self, db, row_norm_bound=None, reserve: int = 6 << 30):
        """The prefilter's image of a resident matrix (ops.pf_build_image; the fp32 rows stay for the exact re-scoring): fp16 rows,
        +256 B per row (MS_PF_F16X2, the default; MS_PF_FORMAT overrides) -- or the split-bf16 image, +512 B per row, when the rows'
        norm bound leaves the range the fp16 image covers -- or None when HBM has no room for it next to `reserve` bytes of
        workspace: the prefiltered search then splits the rows in registers (inner-product modes) or the fp32 scan runs (cosine)."""
        from .. import _lib
        ops = self._ops
        n = int(db.shape[0])
        if n < _lib.PREFILTER_MIN_ROWS or os.environ.get("MERIZO_PF_IMAGE", "1") == "0":
            return None
        bound = float(row_norm_bound) if row_norm_bound is not None else self.row_norm_bound(db)
        fmt = ops.pf_default_format()
        if fmt != ops.PF_BF16X3 and not (2.0 ** -40 <= bound <= 2.0 ** 40):
            fmt = ops.PF_BF16X3
        need = int(_lib.load().ms_pf_image_bytes(n, fmt))
        free, _total = self.torch.cuda.mem_get_info(self.device)
        if need + reserve > free:
            logger.info("no room for the prefilter's image (%d MiB, %d MiB free): splitting rows in registers" % (need >> 20, free >> 20))
            return None
        image = ops.pf_build_image(db, fmt=fmt, row_norm_bound=bound)
        if fmt != ops.PF_BF16X3 and ops.pf_format_is_auto():
            # one or two matrix instructions per 16 dimensions?  Decided once per database by searching 256 of its own rows
            image = ops.pf_choose_format(db, image, bound)
            logger.info("prefilter arithmetic for this database: %s" % ("MS_PF_F16X1" if image.format == ops.PF_F16X1 else "MS_PF_F16X2"))
        return image

    def lazy_pf_image(self, db, row_norm_bound=None):
        """pf_image(db), built by the FIRST batch the prefiltered search serves (more than 64 queries): a CLI run with a handful of
        query domains never pays the extra pass over the rows nor the image's memory."""
        return LazyPfImage(self, db, row_norm_bound)

    def row_norm_bound(self, db) -> float:
        """max |row| over a resident database, a hair up (one HBM pass; the prefiltered search's error bound scales with it)."""
        if db.shape[0] == 0:
            return 1.0
        inv_min = float(self._ops.row_inv_norms(db, 1e-30).min())
        return (1.0 / inv_min) * (1.0 + 1e-6) if inv_min > 0.0 else float("inf")

    def topk_merge(self, scores, idx):
        return self._ops.topk_merge(scores, idx)

    def merge_gathered(self, exchange):
        """Global top-k from the all-gathered per-shard blocks, read in place (ms_topk_merge_strided)."""
        return exchange.merge()

    # -- database residency ----------------------------------------------------------
    STAGE_ROWS = 1 << 19          # 256 MiB pinned staging buffers
    COPY_THREADS = int(os.environ.get("MERIZO_COPY_THREADS", "0")) or min(64, os.cpu_count() or 8)      # (64: 44-48 GB/s streamed on a 128-core host against 39-43 with 32 or 96: profiles/r06_streamed_copy_threads.log)
    # host threads filling a staging buffer (pread / numpy release the GIL while they copy)

    @staticmethod
    def _file_span(src):
        """(file name, byte offset) of a C-contiguous np.memmap (or a row slice of one), else None."""
        if not isinstance(src, np.memmap) or not src.flags.c_contiguous or getattr(src, "filename", None) is None:
            return None
        root = src
        while isinstance(root.base, np.memmap):
            root = root.base
        return str(root.filename), int(root.offset) + (src.ctypes.data - root.ctypes.data)

    def _host_copy(self, dst, src) -> None:
        """dst[:] = src for float32 [rows,128] arrays, split over COPY_THREADS threads.  A memmap source is read with
        pread straight from the file into the pinned destination (the kernel copies out of the page cache in large pieces:
        no page faults, ~3x the rate of copying through the mapping); anything else is copied with numpy."""
        rows = src.shape[0]
        nbytes = rows * src.shape[1] * 4
        span = self._file_span(src)
        if rows < (1 << 14) or self.COPY_THREADS <= 1:
            np.copyto(dst, src)
            return
        pool = getattr(self, "_copy_pool", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._copy_pool = ThreadPoolExecutor(max_workers=self.COPY_THREADS)
        if span is not None:
            fname, off = span
            fd = os.open(fname, os.O_RDONLY)
            try:
                flat = memoryview(dst.reshape(-1).view(np.uint8))
                step = max(1 << 21, (nbytes + self.COPY_THREADS - 1) // self.COPY_THREADS)
                step = (step + 4095) // 4096 * 4096

                def read(a):
                    b = min(nbytes, a + step)
                    while a < b:
                        got = os.preadv(fd, [flat[a:b]], off + a)
                        if got <= 0:
                            raise IOError("short read from %s at byte %d" % (fname, off + a))
                        a += got
                list(pool.map(read, range(0, nbytes, step)))
            finally:
                os.close(fd)
            return
        step = (rows + self.COPY_THREADS - 1) // self.COPY_THREADS
        list(pool.map(lambda a: np.copyto(dst[a:a + step], src[a:a + step]), range(0, rows, step)))

    def resident_budget(self, nq: int = 4096, k: int = 64) -> int:
        """Bytes of HBM a resident matrix may take: what is free now minus the scan workspace of a (nq, k) batch over a
        matrix of that size (partial lists, padded queries, the exchange block of a sharded search), the device side of
        the pinned-upload staging and a margin for the allocator and the encoder."""
        from .. import _lib
        free, total = self.torch.cuda.mem_get_info(self.device)
        margin = (2 << 30) + total // 50 + 2 * self.STAGE_ROWS * 512
        rows_est = max(1, min((free - margin) // 512, (1 << 31) - 2))
        with self.torch.cuda.device(self.device):        # (the prefiltered search's workspace covers the plain one's)
            ws = int(_lib.load().ms_ip_topk_prefiltered_workspace_bytes(int(rows_est), max(1, int(nq)), max(1, int(k))))
        ws += 3 * 12 * max(1, int(nq)) * max(1, int(k))            # outputs + PackedExchange blocks
        ws += 4 * int(rows_est)                                     # row_norm_bound's one float per row, right after the upload
        return max(0, int(free - margin - ws))

    def _staging(self, rows: int):
        bufs = getattr(self, "_pinned", None)
        if bufs is None or bufs[0].shape[0] < rows:
            bufs = [self.torch.empty((rows, W.DIM), dtype=self.torch.float32).pin_memory() for _ in range(2)]
            self._pinned = bufs
        return bufs

    def upload_rows(self, matrix, lo: int, hi: int):
        """Rows [lo,hi) of a host float32 [N,128] matrix (np.memmap of `dbfname_IP`, or an array) as ONE
        contiguous device tensor.  The copy goes through two pinned staging buffers on a side stream:
        the host reads chunk c+1 from the page cache / disk while chunk c crosses PCIe."""
        torch = self.torch
        n = hi - lo
        out = torch.empty((n, W.DIM), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        step = min(self.STAGE_ROWS, n)
        bufs = self._staging(step)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))     # `out` may reuse memory the main stream is still reading
        busy = [None, None]
        for c, r0 in enumerate(range(0, n, step)):
            r1 = min(n, r0 + step)
            slot = c & 1
            if busy[slot] is not None:
                busy[slot].synchronize()                    # the staging buffer's previous copy has left it
            self._host_copy(bufs[slot][: r1 - r0].numpy(), matrix[lo + r0: lo + r1])
            with torch.cuda.stream(side):
                out[r0:r1].copy_(bufs[slot][: r1 - r0], non_blocking=True)
                busy[slot] = side.record_event()
        torch.cuda.current_stream(self.device).wait_stream(side)
        side.synchronize()
        return out

    STREAM_SLOTS = 3

    def _stream_buffers(self, slot: int, rows: int):
        """Pinned + device staging buffer number `slot` of the streamed search, at least `rows` rows: allocated ONCE per engine and reused by
        every later call (round 6: pinning two fresh 128 MiB buffers per call was half of an 8M-row search: 19 -> 33 GB/s)."""
        torch = self.torch
        st = getattr(self, "_stream_state", None)
        if st is None:
            st = self._stream_state = {"pinned": [None] * self.STREAM_SLOTS, "dev": [None] * self.STREAM_SLOTS,
                                       "side": torch.cuda.Stream(device=self.device)}
        if st["pinned"][slot] is None or st["pinned"][slot].shape[0] < rows:
            st["pinned"][slot] = torch.empty((rows, W.DIM), dtype=torch.float32).pin_memory()
            with torch.cuda.device(self.device):
                st["dev"][slot] = torch.empty((rows, W.DIM), dtype=torch.float32, device=self.device)
            return st["pinned"][slot], st["dev"][slot], True
        return st["pinned"][slot], st["dev"][slot], False

    def device_blocks(self, blocks):
        """Out-of-core streaming (the reference's db_iterator loop, dbsearch.py:233-243): yields each host block as a device
        tensor.  Three staging slots filled by a BACKGROUND thread (round 6): while the consumer enqueues and runs its scan of block b,
        the thread reads block b + 1 (b + 2) into pinned memory (the host copy releases the GIL) and copies it on a side stream; a device
        buffer is overwritten only after the work the consumer enqueued on it has finished (an event recorded when the consumer asks for
        the next block).  Before: the consumer's own host-side work per block (launches, the running merge) and the host copy of the next
        block took turns on one thread."""
        import queue
        import threading
        torch = self.torch
        main = torch.cuda.current_stream(self.device)
        self._stream_buffers(0, 1)                              # (creates the state)
        st = self._stream_state
        side = st["side"]
        side.wait_stream(main)                                  # the buffers' previous users (an earlier call's scans) are on the main stream
        ready = queue.Queue(maxsize=self.STREAM_SLOTS - 1)      # staged blocks: (slot, rows, copied event) | None (end) | an exception
        free = queue.Queue()                                    # slots handed back: (slot, event after which its device buffer may be overwritten)
        for s_ in range(self.STREAM_SLOTS):
            free.put((s_, None))
        stop = threading.Event()
        copied = [None] * self.STREAM_SLOTS

        def put_ready(item) -> bool:
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def worker():
            try:
                torch.cuda.set_device(self.device)
                for block in blocks:
                    if isinstance(block, torch.Tensor):
                        block = block.numpy()
                    rows = block.shape[0]
                    if rows == 0:
                        continue
                    slot = None
                    while not stop.is_set():
                        try:
                            slot, used = free.get(timeout=0.05)
                            break
                        except queue.Empty:
                            continue
                    if slot is None:
                        return
                    pinned, dev, fresh = self._stream_buffers(slot, rows)
                    if copied[slot] is not None:
                        copied[slot].synchronize()              # the previous copy out of this pinned buffer has left it
                    self._host_copy(pinned[:rows].numpy(), block)
                    with torch.cuda.stream(side):
                        if fresh:
                            side.wait_stream(main)              # a fresh buffer may reuse memory the main stream is still reading
                        if used is not None:
                            side.wait_event(used)
                        dev[:rows].copy_(pinned[:rows], non_blocking=True)
                        copied[slot] = side.record_event()
                    if not put_ready((slot, rows, copied[slot])):
                        return
                put_ready(None)
            except BaseException as exc:                        # (reported to the consumer, which re-raises it)
                put_ready(exc)

        th = threading.Thread(target=worker, name="merizo-stream-stage", daemon=True)
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, rows, ev = item
                main.wait_event(ev)
                yield st["dev"][slot][:rows]                    # the consumer enqueues its scan of this block ...
                free.put((slot, main.record_event()))           # ... and the slot goes back once that work has been enqueued
        finally:
            stop.set()
            th.join(timeout=30.0)
