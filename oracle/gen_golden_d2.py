"""TEST INFRASTRUCTURE -- runs in the build container only (imports the reference from /root/reference).

Golden vectors for the encoder with a LARGE distance-column weight: synthetic_state_dict(seed 0, d2_scale = 1.0), i.e. the
squared-distance column of edge_mlp.0.weight at the scale of the other columns, so that w * d^2 (d^2 reaches 1e4 A^2) drives
the first SiLU deep into saturation (pre-activations of +-1e2 .. +-1e3: the range SURVEY.md 7 warns about; the default
fixtures scale that column by 1/64).  Writes tests/golden/egnn_d2.npz: coordinates and FoldClassNet embeddings.

`long` (round 5): the same network on LONG chains -- random walks of 1000 and 2000 residues (2000 = createdb's truncation length,
makedb.py:68-69), with d2_scale = 1.0 and with the default fixture weights -> tests/golden/egnn_long.npz (the reference materialises
[N^2, 514] floats per layer: 8 GB and about a minute per structure at N = 2000).

    python oracle/gen_golden_d2.py [long]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/merizo_search")
from programs.Foldclass.nndef_fold_egnn_embed import FoldClassNet  # noqa: E402

from merizo_search_amd.foldclass import synthetic as syn, weights as W  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def main():
    torch.manual_seed(0)
    sd = W.synthetic_state_dict(0, d2_scale=1.0)
    net = FoldClassNet(128).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    pdb = np.load(os.path.join(OUT, "pdb_M0.npz"))
    cases = [("M0", pdb["coords"]), ("walk97", syn.random_walk(97, seed=197)), ("walk292", syn.random_walk(292, seed=392))]
    arrays = {"d2_scale": np.float32(1.0)}
    with torch.no_grad():
        for name, coords in cases:
            emb = net(torch.from_numpy(coords).unsqueeze(0))[0].numpy()
            arrays[f"coords_{name}"] = coords
            arrays[f"emb_{name}"] = emb
            print(name, coords.shape, "max |e| = %.4g" % float(np.abs(emb).max()))
    np.savez_compressed(os.path.join(OUT, "egnn_d2.npz"), **arrays)


def main_long():
    arrays = {}
    for tag, d2 in (("d2", 1.0), ("std", None)):
        torch.manual_seed(0)
        sd = W.synthetic_state_dict(0, d2_scale=d2) if d2 is not None else W.synthetic_state_dict(0)
        net = FoldClassNet(128).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        with torch.no_grad():
            for n, seed in ((1000, 1197), (2000, 2197)):
                coords = syn.random_walk(n, seed=seed)
                emb = net(torch.from_numpy(coords).unsqueeze(0))[0].numpy()
                arrays[f"coords_walk{n}"] = coords
                arrays[f"emb_{tag}_walk{n}"] = emb
                print(tag, n, "max |e| = %.4g" % float(np.abs(emb).max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "egnn_long.npz"), **arrays)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        main_long()
    else:
        main()
