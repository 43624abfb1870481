"""Optional TM-align verification hook (mirror of programs/Foldclass/utils.py:75-158).

TM-align is a third-party CPU binary the reference shells out to for every hit; it is not
shipped with the reference snapshot and is outside the accelerated path.  When a binary is
available ($MERIZO_TMALIGN, or `tmalign` / `TMalign` on PATH, or next to this file) the
drivers call it exactly like the reference; otherwise searches run embedding-only
(skip_tmalign) and say so.
"""
from __future__ import annotations

import logging
import os
import re
import shutil
import subprocess
from typing import Optional

logger = logging.getLogger(__name__)

_ALIGNED = re.compile(r"Aligned length=\s*(\d+),\s+RMSD=\s*([0-9.]+),\s+Seq_ID=n_identical/n_aligned=\s*([0-9.]+)")
_TMSCORE = re.compile(r"TM-score=\s*([0-9.]+)")


def find_tmalign() -> Optional[str]:
    cands = [os.environ.get("MERIZO_TMALIGN"), os.path.join(os.path.dirname(os.path.realpath(__file__)), "tmalign"),
             shutil.which("tmalign"), shutil.which("TMalign")]
    for c in cands:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


def extract_tmalign_values(tmalign_output: str, return_alignment: bool = False) -> dict:
    """Parse TM-align's stdout -> {len_ali, rmsd, seq_id, qtm, ttm[, alignment]}.
    Two `TM-score=` lines are expected (normalised by query, then by target)."""
    m = _ALIGNED.search(tmalign_output)
    scores = [float(x) for x in _TMSCORE.findall(tmalign_output)]
    result = {
        "len_ali": int(m.group(1)) if m else None,
        "rmsd": float(m.group(2)) if m else None,
        "seq_id": float(m.group(3)) if m else None,
        "qtm": scores[0],
        "ttm": scores[1],
    }
    if return_alignment:
        start = tmalign_output.find('(":" denotes residue pairs')
        result["alignment"] = tmalign_output[start:].split("\n")[1:4]
    return result


def run_tmalign(structure1_path: str, structure2_path: str, options: Optional[str] = None, keep_pdbs: bool = False,
                binary: Optional[str] = None):
    """Run TM-align on two CA-only PDB files; returns the parsed dict ("" on failure, like the
    reference).  Input files are removed unless keep_pdbs."""
    binary = binary or find_tmalign()
    if binary is None:
        raise FileNotFoundError("no TM-align binary found (set $MERIZO_TMALIGN)")
    cmd = [binary, structure1_path, structure2_path] + ([options] if options else [])
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if proc.returncode != 0:
        logger.error(f"Error running tmalign: {proc.stderr}")
        return ""
    if not keep_pdbs:
        for path in (structure1_path, structure2_path):
            try:
                os.remove(path)
            except OSError as exc:
                logger.error(f"Error deleting structure files: {exc}")
    return extract_tmalign_values(proc.stdout)
