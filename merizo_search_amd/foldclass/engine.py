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
    row_inv_norms(db, eps)                                  -> float32 [n]
    cosine_topk(db, q, k, inv_norm, lengths, qlen, mincov)  -> (scores [nq,k], idx int64 [nq,k])
    ip_topk(db, q, k, row_offset)                           -> (scores [nq,k], idx int64 [nq,k])
    topk_merge(scores [S,nq,k], idx [S,nq,k])               -> (scores [nq,k], idx [nq,k])
"""
from __future__ import annotations

import logging
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

    def row_inv_norms(self, db, eps: float = 1e-8):
        return self._ops.row_inv_norms(db, eps)

    def cosine_topk(self, db, q, k, inv_norm=None, lengths=None, qlen=None, mincov: float = 0.0):
        return self._ops.ip_topk(db, q, k, mode=self._ops.MODE_COSINE_RAW, inv_norm=inv_norm, lengths=lengths,
                                 qlen=qlen, mincov=mincov, workspace=self._ws)

    def ip_topk(self, db, q, k, row_offset: int = 0):
        return self._ops.ip_topk(db, q, k, mode=self._ops.MODE_IP_PRENORM, row_offset=row_offset, workspace=self._ws)

    def topk_merge(self, scores, idx):
        return self._ops.topk_merge(scores, idx)
