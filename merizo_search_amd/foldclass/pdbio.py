"""CA-trace PDB input/output for the Foldclass path.

Mirrors programs/Foldclass/utils.py: read_pdb (:42-72) and write_pdb (:14-39), and the
all-chains / 2000-residue variant inlined in makedb.py:58-69.
"""
from __future__ import annotations

import logging
import os
import sys
import uuid

import numpy as np

from .constants import single_to_three_aa, three_to_single_aa

logger = logging.getLogger(__name__)


def _ca_record(line: str):
    """(xyz, one-letter residue) of an `ATOM ... CA` record, else None."""
    if line[:4] != "ATOM" or line[12:16] != " CA ":
        return None
    xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
    return xyz, three_to_single_aa.get(line[17:20], "X")


def read_pdb(pdbfile: str, pdb_chain: str = "A") -> dict:
    """CA coordinates and sequence of one chain -> {'coords': f32[N,3], 'seq': str, 'name': path}.

    Reference behaviour kept (utils.py:42-72): chain id must be one character (exit 127); only
    lines whose column 22 equals the chain are considered, and that column is inspected before
    any length check, so a line shorter than 22 characters raises IndexError like the reference;
    unknown residue names become 'X'; no CA atoms for the chain -> exit 128; no truncation.
    """
    if len(pdb_chain) != 1:
        logger.error("Invalid chain ID: '%s'" % pdb_chain)
        sys.exit(127)
    xyz, seq = [], []
    with open(pdbfile, "r") as handle:
        for line in handle:
            if line[21] != pdb_chain:
                continue
            rec = _ca_record(line)
            if rec is not None:
                xyz.append(rec[0])
                seq.append(rec[1])
    if not seq:
        logger.error("Chain ID '%s' not present in PDB file %s." % (pdb_chain, pdbfile))
        sys.exit(128)
    return {"coords": np.asarray(xyz, dtype=np.float64).astype(np.float32), "seq": "".join(seq), "name": pdbfile}


def read_pdb_all_chains(pdbfile: str, max_len: int = 2000):
    """createdb's parser (makedb.py:58-69): every CA ATOM record regardless of chain, truncated
    to `max_len` residues.  Returns (coords f32[N,3], seq) -- N may be 0."""
    xyz, seq = [], []
    with open(pdbfile, "r") as handle:
        for line in handle:
            rec = _ca_record(line)
            if rec is not None:
                xyz.append(rec[0])
                seq.append(rec[1])
    coords = np.asarray(xyz, dtype=np.float64).astype(np.float32).reshape(-1, 3)[:max_len]
    return coords, "".join(seq[:max_len])


def write_pdb(tmp: str, coords, sequence: str, name: str | None = None) -> str:
    """Write a CA-only PDB (chain A, residues numbered from 1) into `tmp`; returns the path.

    Record format as utils.py:26-30; the file name is a uuid4 unless `name` is given."""
    assert len(coords) == len(sequence), "Number of coordinates should match number of amino acids"
    filename = os.path.join(tmp, (name if name is not None else str(uuid.uuid4())) + ".pdb")
    with open(filename, "w") as out:
        for num, (xyz, aa) in enumerate(zip(coords, sequence), start=1):
            out.write("ATOM  %5d  CA  %3s A%4d    %8.3f%8.3f%8.3f  1.00  0.00\n"
                      % (num, single_to_three_aa.get(aa), num, xyz[0], xyz[1], xyz[2]))
        out.write("END\n")
    return filename
