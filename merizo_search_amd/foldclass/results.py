"""TSV writers and CLI-side checks (mirror of programs/utils.py).

Column formats follow programs/utils.py:98-142: emb_score / TM fields "{:.4f}", rmsd "{:.2f}",
emb_rank = the hit's key in its per-query dict, metadata written as the raw string.
"""
from __future__ import annotations

import ast
import json
import logging
import os
import sys

logger = logging.getLogger(__name__)

SEARCH_FIELDS = "query,emb_rank,target,emb_score,q_len,t_len,ali_len,seq_id,q_tm,t_tm,max_tm,rmsd,metadata"
EASY_SEARCH_FIELDS = "query,chopping,conf,plddt,emb_rank,target,emb_score,q_len,t_len,ali_len,seq_id,q_tm,t_tm,max_tm,rmsd,metadata"

_HEADER_NAME = {"conf": "dom_conf", "plddt": "dom_plddt"}
_TM_FIELDS = {"ali_len", "seq_id", "q_tm", "t_tm", "max_tm", "rmsd"}


def parse_output_format(format_str: str, expected_str: str):
    """Validate a comma-separated --format list against the allowed names (programs/utils.py:8-21)."""
    wanted = format_str.split(",")
    allowed = expected_str.split(",")
    if not wanted:
        logger.error("No fields found in the provided format string.")
        sys.exit(1)
    for field in wanted:
        if field not in allowed:
            logger.warning(f"Format option '{field}' is not recognized.")
            sys.exit(1)
    return wanted


def check_for_database(db_name: str) -> None:
    """A faiss-layout DB is recognised by <db>.json; otherwise <db>.pt and <db>.index must exist
    (programs/utils.py:23-35)."""
    if os.path.exists(db_name + ".json"):
        return
    for ext in (".pt", ".index"):
        if not os.path.exists(db_name + ext):
            logger.error(f"Cannot find database file {db_name + ext}")
            sys.exit(1)


def _cell(option: str, rank, match: dict) -> str:
    tm = match.get("tmalign_output")
    if option in _TM_FIELDS and tm is None:
        raise TypeError(f"format field '{option}' needs TM-align output, but this hit was not aligned "
                        "(embedding-only search); drop the TM fields from --format")
    if option == "query":
        return match["query"]
    if option == "target":
        return match["target"]
    if option == "chopping":
        return match["dom_str"]
    if option == "conf":
        return "{:.4f}".format(match["dom_conf"])
    if option == "plddt":
        return "{:.4f}".format(match["dom_plddt"])
    if option == "emb_rank":
        return "{}".format(rank)
    if option == "emb_score":
        return "{:.4f}".format(match["score"])
    if option == "q_len":
        return "{}".format(match["q_len"])
    if option == "t_len":
        return "{}".format(match["t_len"])
    if option == "ali_len":
        return "{}".format(tm["len_ali"])
    if option == "seq_id":
        return "{:.4f}".format(tm["seq_id"])
    if option == "q_tm":
        return "{:.4f}".format(tm["qtm"])
    if option == "t_tm":
        return "{:.4f}".format(tm["ttm"])
    if option == "max_tm":
        return "{:.4f}".format(max(tm["qtm"], tm["ttm"]))
    if option == "rmsd":
        return "{:.2f}".format(tm["rmsd"])
    if option == "metadata":
        md = match["metadata"]
        if isinstance(md, dict):
            return "\t".join(md.values())
        return "{}".format(md)
    logger.warning(f"Format option '{option}' is not recognized.")
    sys.exit(1)


def write_search_results(results, output_file: str, format_list, header: bool, metadata_json: bool = False) -> None:
    """results: list (one per query) of dict rank -> hit dict (dbsearch.py:126-138)."""
    known = set(EASY_SEARCH_FIELDS.split(","))
    with open(output_file, "w+") as out:
        if header:
            for option in format_list:
                if option not in known:
                    logger.warning(f"Format option '{option}' is not recognized.")
                    sys.exit(1)
            out.write("\t".join(_HEADER_NAME.get(o, o) for o in format_list).rstrip() + "\n")
        for per_query in results:
            for rank, match in per_query.items():
                out.write("\t".join(_cell(o, rank, match) for o in format_list) + "\n")
    if metadata_json and len(results) > 0:
        md = [ast.literal_eval(hit["metadata"]) for per_query in results if per_query
              for hit in per_query.values() if hit["metadata"] != "{ }"]
        path = output_file + ".hit_metadata.json"
        with open(path, "w") as handle:
            json.dump(md, handle)
        logger.info("Metadata for hits written to " + path)


def write_segment_results(results, output_file: str, header: bool) -> None:
    """`_segment.tsv` pass-through (programs/utils.py:161-176)."""
    with open(output_file, "w+") as out:
        if header:
            out.write("filename\tnres\tnres_dom\tnres_ndr\tndom\tpIoU\truntime\tresult\n")
        for res in results:
            out.write("{}\t{}\t{}\t{}\t{}\t{:.4f}\t{:.4f}\t{}\n".format(
                os.path.basename(res["name"]).replace(".pdb", ""), int(res["length"]), int(res["nres_domain"]),
                int(res["nres_non_domain"]), int(res["num_domains"]), res["conf"], res["time"], res["dom_str"]))


def write_all_dom_search_results(results, output_file: str, header: bool) -> None:
    if results is None:
        return
    with open(output_file, "w+") as out:
        if header:
            out.write("query_chain\tnqd\thit_chain\tnhd\tmatch_category\tmatch_info\thit_metadata\n")
        for res in results:
            out.write("\t".join(str(a) for a in res) + "\n")
