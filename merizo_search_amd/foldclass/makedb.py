"""createdb: embed a directory of PDB files into a Foldclass database.

Mirror of programs/Foldclass/makedb.py:34-94 (run_createdb): every `*.pdb` in the directory in
sorted order, all chains' CA atoms, truncated to 2000 residues, files without CA atoms skipped
with a warning; output `<out_db>.pt` (raw embeddings) + `<out_db>.index` (pickled list of
(path, coords, seq)).  The reference embeds one structure per network call; here the whole
directory goes through ragged GPU launches (under torchrun: data-parallel over the ranks, rank 0
writes).  ``layout="faiss"`` additionally writes the
faiss layout (normalised matrix + names / sequence / CA files), which the reference has no
writer for.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import numpy as np

from . import sharded
from .dbutil import NAME_WIDTH, write_faiss_db, write_pt_db
from .network import network_setup
from .pdbio import read_pdb_all_chains


def _faiss_layout_names(paths) -> list:
    """File stems as the 32-byte name records of the faiss layout (dbutil.py:107-108).  A longer stem
    is cut to 32 bytes with a warning; a cut that makes two entries indistinguishable is an error."""
    stems = [os.path.basename(p).replace(".pdb", "") for p in paths]
    short = [s[:NAME_WIDTH] for s in stems]
    cut = [s for s in stems if len(s) > NAME_WIDTH]
    if cut:
        logging.warning(f"{len(cut)} entry names exceed the {NAME_WIDTH}-byte name record of the faiss layout and were "
                        f"truncated (first: {cut[0]} -> {cut[0][:NAME_WIDTH]}).")
        if len(set(short)) != len(set(stems)):
            seen, clash = {}, None
            for full, sh in zip(stems, short):
                if seen.setdefault(sh, full) != full:
                    clash = (seen[sh], full)
                    break
            raise ValueError(f"entry names {clash[0]} and {clash[1]} are identical in their first {NAME_WIDTH} bytes: "
                             "rename the files, the faiss layout cannot tell them apart")
    return short


def run_createdb(pdb_files: str, out_db: str, device: str = "cuda", network=None, layout: str = "pt",
                 weights_path: Optional[str] = None) -> int:
    paths = sorted(os.path.join(pdb_files, f) for f in os.listdir(pdb_files) if f.endswith(".pdb"))
    logging.info(f"{len(paths)} PDB files found in model directory. Will generate Foldclass database..")
    if network is None:
        network, device = network_setup(threads=-1, device=device, weights_path=weights_path)
    names, coords, seqs = [], [], []
    for path in paths:
        ca, seq = read_pdb_all_chains(path, max_len=2000)
        if len(ca) == 0 or len(seq) == 0:
            logging.warning("No CA atoms read from PDB file " + path + "; skipping.")
            continue
        names.append(path); coords.append(ca); seqs.append(seq)
    logging.info(f"Output database contains {len(names)} PDBs.")
    if not names:
        raise RuntimeError("no structures to embed")       # torch.cat([]) fails in the reference too
    emb = sharded.embed_distributed(network, coords)     # ragged launches; split over the ranks under torchrun
    if sharded.rank_world()[0] != 0:
        return len(names)                                 # rank 0 writes the files
    if layout in ("pt", "both"):
        write_pt_db(out_db, emb.cpu().numpy(), names, coords, seqs)
        logging.info(f"Saved Foldclass database to {out_db}.pt")
        logging.info(f"Saved Foldclass index file to {out_db}.index")
    if layout in ("faiss", "both"):
        normed = network.engine.normalized(emb, 1e-12).cpu().numpy()
        short = _faiss_layout_names(names)
        write_faiss_db(out_db, normed, short, seqs, coords)
        logging.info(f"Saved faiss-layout database to {out_db}.json")
    return len(names)
