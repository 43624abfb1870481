"""torch-CPU expression of the reference's three hot loops, for the `cpu_baseline` leg of bench.py.

TEST / BASELINE INFRASTRUCTURE ONLY (like everything under oracle/): nothing under
merizo_search_amd/ imports this module.  The reference itself is Python and cannot travel to the
GPU box, so BASELINE.md section 4 asks for the SAME torch ops in the SAME shapes, written here and
pinned against the reference-derived golden vectors by tests/test_oracle_golden.py:

  pt_path_search     dbsearch.py:75-81    per query: F.cosine_similarity(db, q) * mask -> torch.topk
  faiss_path_search  dbsearch.py:213-248  F.normalize(Q); per 262,144-row block: Q @ block^T -> topk ->
                                          `I += i0` -> merge into the running top-k (IndexFlat + ResultHeap)
  egnn_forward       my_egnn_nocoords.py:44-74 + nndef_fold_egnn_embed.py:50-62, batch = 1: the literal
                     arithmetic (materialises [N,N,257] and [N,N,514])
"""
from __future__ import annotations

import time
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def pt_path_search(db: torch.Tensor, lengths: torch.Tensor, q: torch.Tensor, qlen: Sequence[float], mincov: float, k: int):
    """One call per query, like the loop at dbsearch.py:531-546 -> (scores [nq,k], idx [nq,k])."""
    out_s, out_i = [], []
    for j in range(q.shape[0]):
        mask = (float(qlen[j]) >= lengths * mincov).float()                       # :76
        scores = F.cosine_similarity(db, q[j:j + 1], dim=-1) * mask                # :78
        top = torch.topk(scores, k, dim=0)                                         # :79
        out_s.append(top.values); out_i.append(top.indices)
    return torch.stack(out_s), torch.stack(out_i)


def faiss_path_search(db: torch.Tensor, q: torch.Tensor, k: int, block: int = 262144):
    """knn_exact_faiss with torch standing in for faiss.IndexFlat / ResultHeap (faiss is absent)."""
    xq = F.normalize(q)                                                            # :303-304
    best_s = best_i = None
    for i0 in range(0, db.shape[0], block):                                        # db_iterator, :233
        xb = db[i0:i0 + block]
        top = torch.topk(xq @ xb.T, min(k, xb.shape[0]), dim=1)                    # index.add + index.search, :236-237
        idx = top.indices + i0                                                     # I += i0, :238
        if best_s is None:
            best_s, best_i = top.values, idx
        else:                                                                      # rh.add_result, :240
            cs, ci = torch.cat([best_s, top.values], 1), torch.cat([best_i, idx], 1)
            keep = torch.topk(cs, k, dim=1)
            best_s, best_i = keep.values, torch.gather(ci, 1, keep.indices)
    return best_s, best_i                                                          # rh.finalize, :245


def _layer(sd: Dict[str, torch.Tensor], layer: int, feats: torch.Tensor, coors: torch.Tensor) -> torch.Tensor:
    p = lambda name: sd[f"encode_ca_egnn.{layer}.{name}"]
    n = feats.shape[1]
    rel = coors[:, :, None, :] - coors[:, None, :, :]                              # :48
    dist = torch.linalg.norm(rel, dim=-1, keepdim=True)                            # :49
    fi = feats[:, :, None, :].expand(-1, n, n, -1)
    fj = feats[:, None, :, :].expand(-1, n, n, -1)
    edge_in = torch.cat((fi, fj, dist * dist), dim=-1)                             # :58   [1,N,N,257]
    m = F.silu(F.linear(edge_in, p("edge_mlp.0.weight"), p("edge_mlp.0.bias")))    # :19-20 [1,N,N,514]
    m = F.silu(F.linear(m, p("edge_mlp.2.weight"), p("edge_mlp.2.bias")))          # :21-22 [1,N,N,256]
    m = m * torch.sigmoid(F.linear(m, p("edge_gate.0.weight"), p("edge_gate.0.bias")))   # :64
    m_i = m.sum(dim=-2)                                                            # :69
    h = F.silu(F.linear(torch.cat((feats, m_i), dim=-1), p("node_mlp.0.weight"), p("node_mlp.0.bias")))
    return F.linear(h, p("node_mlp.2.weight"), p("node_mlp.2.bias")) + feats       # :71-72


def egnn_forward(sd: Dict[str, torch.Tensor], coords: torch.Tensor) -> torch.Tensor:
    """FoldClassNet(128).forward for one structure: coords float32 [1,N,3] -> [1,128]."""
    n = coords.shape[1]
    feats = sd["posenc_as.pe"].reshape(1, -1, 128)[:, :n, :]
    with torch.no_grad():
        for layer in range(2):
            feats = _layer(sd, layer, feats, coords)
        return feats.mean(dim=1)


def state_dict_tensors(sd_numpy: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd_numpy.items()}


# ------------------------------------------------------------------ timing legs (bench.py) ----
def time_search_legs(db_unit: np.ndarray, q_unit: np.ndarray, k: int, n_total: int, budget_s: float = 8.0) -> dict:
    """Both search shapes on the first rows of the bench's database, all host threads (torch default)
    and one thread; rows are capped so that each leg stays within ~budget_s; q/s scaled linearly in rows."""
    out = {"threads": torch.get_num_threads()}
    db = torch.from_numpy(db_unit)
    q = torch.from_numpy(q_unit)
    lengths = torch.full((db.shape[0],), 100.0)
    n = db.shape[0]

    def leg(fn, nq_used):
        t = time.perf_counter(); fn(); dt = time.perf_counter() - t
        return {"queries_per_s": nq_used / (dt * n_total / n), "sample": "%d of %d rows x %d queries, %.2f s" % (n, n_total, nq_used, dt)}

    nq1 = min(q.shape[0], 4)
    out["pt_path_per_query"] = leg(lambda: pt_path_search(db, lengths, q[:nq1], [100.0] * nq1, 0.7, k), nq1)
    out["faiss_path_blockwise_262144"] = leg(lambda: faiss_path_search(db, q, k), q.shape[0])
    out["faiss_path_blockwise_262144_nq1"] = leg(lambda: [faiss_path_search(db, q[j:j + 1], k) for j in range(nq1)], nq1)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        sub = db[: max(1, n // 8)]
        t = time.perf_counter(); faiss_path_search(sub, q, k); dt = time.perf_counter() - t
        out["faiss_path_blockwise_262144_1thread"] = {"queries_per_s": q.shape[0] / (dt * n_total / sub.shape[0]),
                                                      "sample": "%d of %d rows x %d queries, %.2f s" % (sub.shape[0], n_total, q.shape[0], dt)}
    finally:
        torch.set_num_threads(threads)
    return out


def time_egnn_leg(sd_numpy: Dict[str, np.ndarray], coords_list: List[np.ndarray], budget_s: float = 10.0) -> dict:
    """The reference's batch = 1 embedding loop (dbsearch.py:288-301) over structures of the bench's
    embed set until ~budget_s have passed; embeds/s extrapolated by sum N^2 to the whole set."""
    sd = state_dict_tensors(sd_numpy)
    done_sq, t0, count = 0, time.perf_counter(), 0
    for c in coords_list:
        egnn_forward(sd, torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))[None])
        done_sq += len(c) ** 2
        count += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    total_sq = sum(len(c) ** 2 for c in coords_list)
    return {"embeds_per_s": len(coords_list) / (dt * total_sq / done_sq), "threads": torch.get_num_threads(),
            "sample": "%d of %d structures (batch = 1 loop), %.2f s, scaled by sum N^2" % (count, len(coords_list), dt)}
