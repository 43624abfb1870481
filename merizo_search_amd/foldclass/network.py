"""network_setup / the encoder callable (mirror of programs/Foldclass/dbsearch.py:35-45).

The reference returns a torch module that is called once per structure,
``network(x: float32[1,N,3]) -> float32[1,128]``.  ``FoldClassEncoder`` keeps that call
signature and adds ``embed_many`` (one ragged launch for a whole batch), which the drivers
in this package use.
"""
from __future__ import annotations

import logging
import os
from typing import Optional, Sequence

import numpy as np

from . import weights as W
from .engine import HipEngine, resolve_device

logger = logging.getLogger(__name__)

WEIGHTS_ENV = "MERIZO_FOLDCLASS_WEIGHTS"
WEIGHTS_NAME = "FINAL_foldclass_model.pt"


def find_checkpoint(explicit: Optional[str] = None) -> Optional[str]:
    """Checkpoint search order: explicit path, $MERIZO_FOLDCLASS_WEIGHTS, next to this file
    (where the reference keeps FINAL_foldclass_model.pt, dbsearch.py:42-43)."""
    for cand in (explicit, os.environ.get(WEIGHTS_ENV), os.path.join(os.path.dirname(os.path.realpath(__file__)), WEIGHTS_NAME)):
        if cand and os.path.exists(cand):
            return cand
    return None


class FoldClassEncoder:
    """Callable stand-in for FoldClassNet(128).eval() on the GPU."""

    def __init__(self, engine):
        self.engine = engine
        self.width = W.DIM

    def __call__(self, x):
        """x: float32 [1,N,3] (tensor or array) -> float32 [1,128] tensor on the device."""
        arr = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        if arr.ndim != 3 or arr.shape[0] != 1 or arr.shape[2] != 3:
            raise ValueError(f"expected coordinates of shape [1,N,3], got {arr.shape}")
        return self.engine.embed([arr[0]])

    def embed_many(self, coords_list: Sequence[np.ndarray]):
        return self.engine.embed(coords_list)

    def eval(self):
        return self

    def to(self, _device):
        return self


def network_setup(threads: int = -1, device="cuda", weights_path: Optional[str] = None, engine=None,
                  allow_synthetic: bool = False):
    """-> (network, device).  `threads` is accepted for signature parity (dbsearch.py:35-37)
    and ignored: nothing on this path runs on CPU threads.

    Weights: a real checkpoint if one is found (find_checkpoint); otherwise, only when
    ``allow_synthetic`` (tests / benchmarks), the seeded synthetic weights.  Without either the
    reference would fail in torch.load; so do we."""
    if engine is None:
        engine = HipEngine(resolve_device(device))
    ckpt = find_checkpoint(weights_path)
    if ckpt is not None:
        sd = W.load_checkpoint(ckpt)
        logger.info("Foldclass weights: %s" % ckpt)
    elif allow_synthetic or os.environ.get("MERIZO_ALLOW_SYNTHETIC_WEIGHTS") == "1":
        sd = W.synthetic_state_dict(0)
        logger.warning("Foldclass checkpoint not found; using SYNTHETIC seeded weights (results are not biological).")
    else:
        raise FileNotFoundError(f"{WEIGHTS_NAME} not found (set ${WEIGHTS_ENV} or pass weights_path)")
    engine.load_weights(sd)
    return FoldClassEncoder(engine), getattr(engine, "device", device)
