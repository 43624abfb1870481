"""Foldclass encoder weights: canonical blob layout, checkpoint loading, synthetic weights.

The encoder's parameters are the reference ``FoldClassNet(128)`` state_dict
(reference: programs/Foldclass/nndef_fold_egnn_embed.py:34-48, my_egnn_nocoords.py:11-37):
two EGNN layers ``encode_ca_egnn.{0,1}.*`` plus the persistent buffer ``posenc_as.pe``.
The HIP library and the CPU oracle both consume one flat fp32 "blob" per layer, the
state_dict tensors flattened row-major in state_dict order (``LAYER_SPEC`` below).

``FINAL_foldclass_model.pt`` is not shipped with the reference snapshot, so tests, smoke and
bench use ``synthetic_state_dict`` (seeded, documented scales); ``load_checkpoint`` reads a
real checkpoint when one is available.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

DIM = 128          # FoldClassNet(128): reference dbsearch.py:40
M_DIM = 256        # m_dim = width * 2: nndef_fold_egnn_embed.py:46
EDGE_IN = 2 * DIM + 1      # my_egnn_nocoords.py:14
EDGE_HID = 2 * EDGE_IN     # my_egnn_nocoords.py:19
NODE_IN = DIM + M_DIM      # my_egnn_nocoords.py:31
NODE_HID = 2 * DIM
MAX_LEN = 3000     # PositionalEncoder(max_len=3000): nndef_fold_egnn_embed.py:12
N_LAYERS = 2       # nndef_fold_egnn_embed.py:45

# (state_dict suffix, shape) in state_dict order
LAYER_SPEC: Tuple[Tuple[str, Tuple[int, ...]], ...] = (
    ("edge_mlp.0.weight", (EDGE_HID, EDGE_IN)),
    ("edge_mlp.0.bias", (EDGE_HID,)),
    ("edge_mlp.2.weight", (M_DIM, EDGE_HID)),
    ("edge_mlp.2.bias", (M_DIM,)),
    ("edge_gate.0.weight", (1, M_DIM)),
    ("edge_gate.0.bias", (1,)),
    ("node_mlp.0.weight", (NODE_HID, NODE_IN)),
    ("node_mlp.0.bias", (NODE_HID,)),
    ("node_mlp.2.weight", (DIM, NODE_HID)),
    ("node_mlp.2.bias", (DIM,)),
)
LAYER_FLOATS = sum(int(np.prod(s)) for _, s in LAYER_SPEC)   # 396165
assert LAYER_FLOATS == 396165
PE_KEY = "posenc_as.pe"


def layer_key(layer: int, suffix: str) -> str:
    return f"encode_ca_egnn.{layer}.{suffix}"


def positional_table(max_len: int = MAX_LEN, d_model: int = DIM) -> np.ndarray:
    """The fixed sinusoid node-feature table, float32 [max_len, d_model].

    The table is DATA: the reference builds it with fp32 torch CPU ops
    (nndef_fold_egnn_embed.py:14-19) and stores it in the checkpoint; a numpy/fp64
    regeneration differs by up to 2e-4.  When no checkpoint supplies ``posenc_as.pe`` we
    therefore build it with the same torch fp32 op sequence: exp(arange(0,d,2) * (-ln 1e4 / d)),
    sin/cos(position * div_term) on even/odd channels.
    """
    import torch

    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    table = torch.zeros(max_len, d_model)
    table[:, 0::2] = torch.sin(position * div_term)
    table[:, 1::2] = torch.cos(position * div_term)
    return table.numpy().copy()


def synthetic_state_dict(seed: int = 0, d2_scale: float = 1.0 / 64.0) -> Dict[str, np.ndarray]:
    """Deterministic stand-in weights with the reference's names and shapes.

    Linear weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2).  The squared-distance column of
    edge_mlp.0.weight is scaled by ``d2_scale`` so that w * d^2 (d^2 reaches 1e4 A^2) stays
    in the range where SiLU is non-linear, i.e. embeddings depend on geometry the way a
    trained network's do.  (The reference's own init, std 1e-3, leaves everything
    bias-dominated: my_egnn_nocoords.py:39-42.)
    """
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for layer in range(N_LAYERS):
        for suffix, shape in LAYER_SPEC:
            if suffix.endswith("weight"):
                fan_in = shape[-1]
                w = rng.standard_normal(shape).astype(np.float32) / np.float32(math.sqrt(fan_in))
                if suffix == "edge_mlp.0.weight":
                    w[:, 2 * DIM] *= np.float32(d2_scale)
                sd[layer_key(layer, suffix)] = w.astype(np.float32)
            else:
                sd[layer_key(layer, suffix)] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
    sd[PE_KEY] = positional_table()[None, :, :]
    return sd


def pack_state_dict(sd: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """state_dict -> (weights float32 [N_LAYERS*LAYER_FLOATS], pe float32 [max_len, DIM]).

    Mirrors ``load_state_dict(..., strict=False)`` (reference dbsearch.py:43): extra keys are
    ignored; a missing ``posenc_as.pe`` falls back to the regenerated table; missing layer
    tensors are an error (strict=False would silently keep random init, which is never wanted).
    """
    chunks = []
    for layer in range(N_LAYERS):
        for suffix, shape in LAYER_SPEC:
            key = layer_key(layer, suffix)
            if key not in sd:
                raise KeyError(f"checkpoint is missing {key}")
            t = np.asarray(sd[key], dtype=np.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{key}: expected shape {shape}, got {tuple(t.shape)}")
            chunks.append(np.ascontiguousarray(t).reshape(-1))
    weights = np.concatenate(chunks)
    assert weights.size == N_LAYERS * LAYER_FLOATS
    if PE_KEY in sd:
        pe = np.asarray(sd[PE_KEY], dtype=np.float32).reshape(-1, DIM)
    else:
        pe = positional_table()
    return weights, np.ascontiguousarray(pe)


def unpack_weights(weights: np.ndarray, pe: np.ndarray | None = None) -> Dict[str, np.ndarray]:
    """Inverse of ``pack_state_dict``."""
    sd: Dict[str, np.ndarray] = {}
    off = 0
    for layer in range(N_LAYERS):
        for suffix, shape in LAYER_SPEC:
            n = int(np.prod(shape))
            sd[layer_key(layer, suffix)] = np.asarray(weights[off:off + n], dtype=np.float32).reshape(shape).copy()
            off += n
    if pe is not None:
        sd[PE_KEY] = np.asarray(pe, dtype=np.float32).reshape(1, -1, DIM)
    return sd


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """Read a Foldclass checkpoint (``FINAL_foldclass_model.pt``: a torch state_dict)."""
    import torch

    sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}
