"""ctypes front-end of the CPU parity oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under merizo_search_amd/ imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile oracle.c with gcc (a few seconds)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_layer_floats.restype = ctypes.c_int
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def num_threads() -> int:
    """OpenMP threads the oracle's parallel loops use."""
    return int(lib().orc_num_threads())


def layer_floats() -> int:
    return int(lib().orc_layer_floats())


def egnn_embed(weights, pe, coords_list, return_layers: bool = False):
    """FoldClassNet forward for a list of [N,3] coordinate arrays -> float32 [B,128]."""
    L = lib()
    weights, wp = _f32(weights)
    pe, pp = _f32(np.asarray(pe).reshape(-1, 128))
    offsets = np.zeros(len(coords_list) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([len(c) for c in coords_list])
    coords, cp = _f32(np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1, 3) for c in coords_list], axis=0))
    out = np.zeros((len(coords_list), 128), dtype=np.float32)
    layers = np.zeros((2, int(offsets[-1]), 128), dtype=np.float32) if return_layers else None
    rc = L.orc_egnn_embed(wp, pp, ctypes.c_int(pe.shape[0]), cp, offsets.ctypes.data_as(_i32p),
                          ctypes.c_int(len(coords_list)), out.ctypes.data_as(_f32p),
                          layers.ctypes.data_as(_f32p) if return_layers else None)
    if rc != 0:
        raise RuntimeError(f"orc_egnn_embed failed: {rc}")
    return (out, layers) if return_layers else out


def l2_normalize_rows(x, eps: float):
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    n, d = x.shape
    lib().orc_l2_normalize_rows(x.ctypes.data_as(_f32p), ctypes.c_int64(n), ctypes.c_int(d), ctypes.c_float(eps))
    return x


def cosine_topk(db, q, k, lengths=None, qlen=None, mincov: float = 0.0, row_offset: int = 0):
    """search_query_against_db for a batch of queries -> (scores f32[nq,k], idx i64[nq,k])."""
    db, dbp = _f32(db)
    q, qp = _f32(np.atleast_2d(q))
    n, d = db.shape
    nq = q.shape[0]
    out_s = np.zeros((nq, k), dtype=np.float32)
    out_i = np.zeros((nq, k), dtype=np.int64)
    if lengths is not None:
        lengths, lp = _f32(lengths)
        qlen, qlp = _f32(np.atleast_1d(qlen))
    else:
        lp = qlp = None
    rc = lib().orc_cosine_topk(dbp, ctypes.c_int64(n), ctypes.c_int(d), ctypes.c_int64(row_offset), qp,
                               ctypes.c_int(nq), ctypes.c_int(k), lp, qlp, ctypes.c_float(mincov),
                               out_s.ctypes.data_as(_f32p), out_i.ctypes.data_as(_i64p))
    if rc == -3:
        raise RuntimeError("selected index k out of range")   # what torch.topk raises for k > n
    if rc != 0:
        raise RuntimeError(f"orc_cosine_topk failed: {rc}")
    return out_s, out_i


def ip_topk(db, q, k, row_offset: int = 0, order: int = 1):
    db, dbp = _f32(db)
    q, qp = _f32(np.atleast_2d(q))
    n, d = db.shape if db.ndim == 2 else (0, q.shape[1])
    nq = q.shape[0]
    out_s = np.zeros((nq, k), dtype=np.float32)
    out_i = np.zeros((nq, k), dtype=np.int64)
    lib().orc_ip_topk(dbp, ctypes.c_int64(n), ctypes.c_int(d), ctypes.c_int64(row_offset), qp, ctypes.c_int(nq),
                      ctypes.c_int(k), ctypes.c_int(order), out_s.ctypes.data_as(_f32p), out_i.ctypes.data_as(_i64p))
    return out_s, out_i


def topk_merge(scores, idx):
    scores, sp = _f32(scores)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    S, nq, k = scores.shape
    out_s = np.zeros((nq, k), dtype=np.float32)
    out_i = np.zeros((nq, k), dtype=np.int64)
    lib().orc_topk_merge(sp, idx.ctypes.data_as(_i64p), ctypes.c_int(S), ctypes.c_int(nq), ctypes.c_int(k),
                         out_s.ctypes.data_as(_f32p), out_i.ctypes.data_as(_i64p))
    return out_s, out_i


def knn_exact_blockwise(db, q, k, block: int = 262144, order: int = 1):
    """knn_exact_faiss over db_iterator(db, block) -> (D f32[nq,k], I i64[nq,k])."""
    db, dbp = _f32(db)
    q, qp = _f32(np.atleast_2d(q))
    n, d = db.shape
    nq = q.shape[0]
    out_s = np.zeros((nq, k), dtype=np.float32)
    out_i = np.zeros((nq, k), dtype=np.int64)
    lib().orc_knn_exact_blockwise(dbp, ctypes.c_int64(n), ctypes.c_int(d), qp, ctypes.c_int(nq), ctypes.c_int(k),
                                  ctypes.c_int64(block), ctypes.c_int(order), out_s.ctypes.data_as(_f32p),
                                  out_i.ctypes.data_as(_i64p))
    return out_s, out_i
