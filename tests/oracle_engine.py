"""An engine with HipEngine's interface backed by the CPU oracle -- TEST INFRASTRUCTURE.

Lets the CPU test-suite exercise the host drivers (database layouts, result assembly, TSV
writers, CLI plumbing, sharding logic) on machines without a GPU.  It lives under tests/ on
purpose: the product package has no CPU path."""
import numpy as np
import torch

from merizo_search_amd.foldclass import weights as W
from oracle import oracle as orc


class OracleEngine:
    name = "oracle"
    device = torch.device("cpu")
    torch = torch

    def __init__(self, state_dict=None):
        self._sd = state_dict
        self._packed = None

    def load_weights(self, state_dict):
        self._sd = state_dict
        self._packed = None

    def embed(self, coords_list, max_batch_sq=None):
        if self._packed is None:
            self._packed = W.pack_state_dict(self._sd)
        weights, pe = self._packed
        return torch.from_numpy(orc.egnn_embed(weights, pe, [np.asarray(c, np.float32) for c in coords_list]))

    def to_device(self, array):
        return array.contiguous() if isinstance(array, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(array))

    def normalize_(self, x, eps=1e-12):
        x.copy_(torch.from_numpy(orc.l2_normalize_rows(x.numpy(), eps)))
        return x

    def normalized(self, x, eps=1e-12):
        return torch.from_numpy(orc.l2_normalize_rows(x.numpy(), eps))

    def row_inv_norms(self, db, eps=1e-8):
        n = np.sqrt((db.numpy().astype(np.float32) ** 2).sum(1, dtype=np.float32))
        return torch.from_numpy((1.0 / np.maximum(n, eps)).astype(np.float32))

    def cosine_rows(self, db):
        return db              # the oracle restates the reference on the raw rows

    def cosine_topk(self, db, q, k, lengths=None, qlen=None, mincov=0.0, row_offset=0):
        s, i = orc.cosine_topk(db.numpy(), q.numpy(), k, None if lengths is None else lengths.numpy(),
                               None if qlen is None else qlen.numpy(), mincov, row_offset=row_offset)
        return torch.from_numpy(s), torch.from_numpy(i)

    def ip_topk(self, db, q, k, row_offset=0, normalize_queries=False):
        qn = orc.l2_normalize_rows(q.numpy(), 1e-12) if normalize_queries else q.numpy()
        s, i = orc.ip_topk(db.numpy(), qn, k, row_offset=row_offset, order=1)
        return torch.from_numpy(s), torch.from_numpy(i)

    def topk_merge(self, scores, idx):
        s, i = orc.topk_merge(scores.numpy(), idx.numpy())
        return torch.from_numpy(s), torch.from_numpy(i)


    def merge_gathered(self, exchange):
        return exchange.merge(merge_fn=self.topk_merge)

    def resident_budget(self, nq=4096, k=64):
        return getattr(self, "budget", 1 << 62)

    def upload_rows(self, matrix, lo, hi):
        return torch.from_numpy(np.array(matrix[lo:hi], dtype=np.float32))

    def device_blocks(self, blocks):
        for b in blocks:
            yield b if isinstance(b, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(b))


def oracle_network(seed=0):
    from merizo_search_amd.foldclass.network import FoldClassEncoder
    eng = OracleEngine(W.synthetic_state_dict(seed))
    return FoldClassEncoder(eng)
