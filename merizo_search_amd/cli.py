"""Command line: `search`, `easy-search`, `createdb` with the reference's arguments.

Mirror of merizo_search/merizo.py for the Foldclass path (search :126-226, easy_search :229-407,
createdb :102-123).  `segment` (the Merizo IPA network) is out of scope: easy-search takes the
chopping as an input (--chopping / --segment_tsv) instead of predicting it.

    python -m merizo_search_amd.cli search  <pdb...> <db_name> <output> <tmp> [-d cuda] [-k 10] ...
    python -m merizo_search_amd.cli createdb <input_dir> <out_db> [-d cuda] [--layout pt|faiss|both]
    python -m merizo_search_amd.cli easy-search <pdb...> <db_name> <output> <tmp> --chopping "71-189,190-290"

Several GPUs of one node: start one process per GPU with torchrun,

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m merizo_search_amd.cli search ...

Every rank then uses GPU LOCAL_RANK (whatever -d says), holds and scans 1/N of the database rows, the
per-shard top-k lists are exchanged with one RCCL all-gather, and rank 0 writes the output files
(foldclass/sharded.py; replaces the reference's index_cpu_to_all_gpus, dbsearch.py:228-230).
"""
from __future__ import annotations

import argparse
import logging
import os
import shutil
import sys
import time
import uuid

from .foldclass import chopping as chop
from .foldclass import sharded
from .foldclass.dbsearch import run_dbsearch
from .foldclass.makedb import run_createdb
from .foldclass.results import (EASY_SEARCH_FIELDS, SEARCH_FIELDS, check_for_database, parse_output_format,
                                write_search_results, write_segment_results)

logging.basicConfig(level=logging.INFO, format="%(asctime)s | %(levelname)s | %(message)s")


def munge_tmp_with_uuid(tmp: str) -> str:
    return os.path.join(tmp, str(uuid.uuid4()))


def _add_search_flags(p: argparse.ArgumentParser, default_format: str) -> None:
    p.add_argument("-d", "--device", type=str, default="cuda", help="'cuda' / 'cuda:N' = the MI355X. 'cpu' is refused by this build.")
    p.add_argument("-k", "--topk", type=int, default=1, help="Max number of domain matches to return per query domain.")
    p.add_argument("-t", "--threads", type=int, default=-1, help="Accepted for compatibility; unused.")
    p.add_argument("-s", "--mincos", type=float, default=0.5, help="Minimum cosine similarity of reported hits.")
    p.add_argument("-m", "--mintm", type=float, default=0.5, help="Minimum TM-align score of reported hits.")
    p.add_argument("-c", "--mincov", type=float, default=0.7, help="Minimum coverage of database matches.")
    p.add_argument("-f", "--fastmode", action="store_true", help="TM-align -fast.")
    p.add_argument("--format", type=str, default=default_format, help="Comma-separated output columns.")
    p.add_argument("--output_headers", action="store_true", default=False)
    p.add_argument("--pdb_chain", type=str, dest="pdb_chain", default="A")
    p.add_argument("--search_batchsize", type=int, default=262144)
    p.add_argument("--search_metric", type=str, default="IP")
    p.add_argument("--report_insignificant_hits", action="store_true", default=False)
    p.add_argument("--metadata_json", action="store_true", default=False)
    p.add_argument("--multi_domain_search", action="store_true", default=False,
                   help="Search DB for entries that match all query domains (all query structures are treated as single "
                        "domains coming from one chain).  Needs a TM-align binary ($MERIZO_TMALIGN).")
    p.add_argument("--multi_domain_mode", type=str, default="exhaustive_tmalign", choices=["exhaustive_tmalign"],
                   help="If --multi_domain_search is used, specifies the multi-domain search mode. Currently only "
                        "'exhaustive_tmalign' is supported.")
    p.add_argument("--skip_tmalign", action="store_true", default=False,
                   help="Embedding-only search (automatic when no TM-align binary is found).")
    p.add_argument("--weights", type=str, default=None, help="Path to FINAL_foldclass_model.pt.")


# The reference's easy-search also takes the Merizo segmenter's options (merizo.py:262-286).  The segmenter is out of scope
# here (the chopping is an input), but a reference command line must still parse: every one of them is accepted with the
# reference's type and default, and reported as ignored when the user set it.
_SEGMENTER_FLAGS = (
    ("--merizo_output", dict(type=str, default=os.environ.get("PWD", "."))),
    ("--save_pdf", dict(action="store_true", default=False)),
    ("--save_pdb", dict(action="store_true", default=False)),
    ("--save_domains", dict(action="store_true", default=False)),
    ("--save_fasta", dict(action="store_true", default=False)),
    ("--conf_filter", dict(type=float, default=None)),
    ("--plddt_filter", dict(type=float, default=None)),
    ("--iterate", dict(action="store_true", default=False)),
    ("--length_conditional_iterate", dict(action="store_true", default=False)),
    ("--max_iterations", dict(type=int, default=3)),
    ("--shuffle_indices", dict(action="store_true", default=False)),
    ("--return_indices", dict(action="store_true", default=False)),
    ("--min_domain_size", dict(type=int, default=50)),
    ("--min_fragment_size", dict(type=int, default=10)),
    ("--domain_ave_size", dict(type=int, default=200)),
    ("--conf_threshold", dict(type=float, default=0.5)),
)


def _add_segmenter_flags(p: argparse.ArgumentParser) -> None:
    g = p.add_argument_group("Merizo segmenter options (accepted for command-line compatibility with the reference, ignored: "
                             "the chopping is an input of this build)")
    for flag, kw in _SEGMENTER_FLAGS:
        g.add_argument(flag, help="Ignored.", **kw)


def _warn_ignored_segmenter_flags(args) -> None:
    given = [flag for flag, kw in _SEGMENTER_FLAGS if getattr(args, flag[2:]) != kw["default"]]
    if given:
        logging.warning("Ignoring the Merizo segmenter option(s) %s: the segmenter is not part of this build, domains come "
                        "from --chopping / --segment_tsv." % ", ".join(given))


def _join_process_group(args) -> None:
    """Under torchrun (WORLD_SIZE > 1): join the process group before the first GPU call and pin this
    rank to its GPU; ranks other than 0 log warnings only."""
    rank, world, device = sharded.init_distributed()
    if world > 1:
        args.device = device
        if rank != 0:
            logging.getLogger().setLevel(logging.WARNING)
        logging.info(f"{world} ranks, one GPU each: database rows are sharded, rank 0 writes the results.")


def _log_command(mode: str) -> None:
    logging.info("Starting %s with command: \n\n%s\n" % (mode, " ".join(f'"{a}"' if " " in a else a for a in sys.argv)))


def _embedding_only_format(fields, skip):
    if not skip:
        return fields
    drop = {"ali_len", "seq_id", "q_tm", "t_tm", "max_tm", "rmsd"}
    kept = [f for f in fields if f not in drop]
    if len(kept) != len(fields):
        logging.warning("TM-align columns dropped from the output (embedding-only search).")
    return kept


def _search_and_write(args, inputs, inputs_are_ca, pdb_chain, fields, tmp):
    from .foldclass import tmalign as tm
    skip = args.skip_tmalign or tm.find_tmalign() is None
    search_output = args.output + "_search.tsv"
    all_output = args.output + "_search_insignificant.tsv"
    for path in (search_output, all_output):
        if os.path.exists(path):
            logging.warning(f"Search output file '{path}' already exists. Results will be overwritten!")
    multi_output = args.output + "_search_multi_dom.tsv"
    if args.multi_domain_search:
        if tm.find_tmalign() is None:
            logging.error("--multi_domain_search aligns every query domain with every candidate target domain and needs a "
                          "TM-align binary (set $MERIZO_TMALIGN).")
            sys.exit(1)
        if os.path.exists(multi_output):
            logging.warning(f"Multi-domain search output file '{multi_output}' already exists. Results will be overwritten!")
    results, all_results = run_dbsearch(
        inputs=inputs, db_name=args.db_name, tmp=tmp, device=args.device, topk=args.topk, fastmode=args.fastmode,
        threads=args.threads, mincos=args.mincos, mintm=args.mintm, mincov=args.mincov, inputs_are_ca=inputs_are_ca,
        pdb_chain=pdb_chain, search_batchsize=args.search_batchsize, search_type=args.search_metric,
        skip_tmalign=skip, weights_path=args.weights)
    if sharded.rank_world()[0] != 0:
        return                                            # every rank searched its shard; rank 0 reports
    fields = _embedding_only_format(fields, skip)
    write_search_results(results=results, output_file=search_output, format_list=fields, header=args.output_headers,
                         metadata_json=args.metadata_json)
    if args.report_insignificant_hits:
        write_search_results(results=all_results, output_file=all_output, format_list=fields, header=args.output_headers,
                             metadata_json=args.metadata_json)
    if args.multi_domain_search:
        # merizo.py:207-222 / :396-411.  `search`: the inputs are single-domain files of ONE chain (the reference
        # passes inputs_from_easy_search=True here, which fails on file names; its stated intent is one chain 'A').
        from .foldclass.multidomain import multi_domain_search
        from .foldclass.results import write_all_dom_search_results
        mda = multi_domain_search(queries=inputs, search_results=results, db_name=args.db_name, tmp_root=tmp, device=args.device,
                                  fastmode=args.fastmode, threads=args.threads, mintm=args.mintm,
                                  inputs_from_easy_search=inputs_are_ca, mode=args.multi_domain_mode, pdb_chain=pdb_chain)
        write_all_dom_search_results(mda, multi_output, args.output_headers)


def search(argv) -> None:
    p = argparse.ArgumentParser(prog="search", description="Search query PDBs against a Foldclass database on the GPU.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("input", type=str, nargs="+")
    p.add_argument("db_name", type=str)
    p.add_argument("output", type=str)
    p.add_argument("tmp", type=str)
    _add_search_flags(p, SEARCH_FIELDS)
    args = p.parse_args(argv)
    _join_process_group(args)
    tmp = munge_tmp_with_uuid(args.tmp)
    _log_command("search")
    check_for_database(args.db_name)
    fields = parse_output_format(args.format, SEARCH_FIELDS)
    t0 = time.time()
    os.makedirs(tmp, exist_ok=True)
    _search_and_write(args, args.input, False, args.pdb_chain, fields, tmp)
    logging.info(f"Finished search in {time.time() - t0} seconds.")
    shutil.rmtree(tmp, ignore_errors=True)


def easy_search(argv) -> None:
    p = argparse.ArgumentParser(prog="easy-search", description="Chop each input chain into domains (chopping supplied) "
                                "and search every domain.", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("input", type=str, nargs="+")
    p.add_argument("db_name", type=str)
    p.add_argument("output", type=str)
    p.add_argument("tmp", type=str)
    _add_search_flags(p, EASY_SEARCH_FIELDS)
    p.add_argument("--chopping", type=str, action="append", default=None,
                   help="Domain chopping of the corresponding input, reference syntax e.g. '71-189,190-290,291-453' "
                        "(repeat the flag once per input).")
    p.add_argument("--segment_tsv", type=str, default=None, help="A reference `_segment.tsv` to take the choppings from.")
    _add_segmenter_flags(p)
    args = p.parse_args(argv)
    _warn_ignored_segmenter_flags(args)
    _join_process_group(args)
    tmp = munge_tmp_with_uuid(args.tmp)
    _log_command("easy-search")
    check_for_database(args.db_name)
    fields = parse_output_format(args.format, EASY_SEARCH_FIELDS)
    chains = args.pdb_chain.rstrip(",").split(",")
    if len(chains) != len(args.input):
        if len(chains) == 1:
            chains = chains * len(args.input)
        else:
            logging.error("Number of specified chain IDs not equal to number of input PDB files.")
            sys.exit(1)
    if args.segment_tsv:
        table = chop.read_segment_tsv(args.segment_tsv)
        choppings = [table.get(os.path.basename(pth).replace(".pdb", "")) for pth in args.input]
    else:
        choppings = args.chopping or []
    if len(choppings) != len(args.input) or any(c is None for c in choppings):
        logging.error("easy-search needs one chopping per input (--chopping ... or --segment_tsv): the Merizo segmenter "
                      "is not part of this build.")
        sys.exit(1)
    t0 = time.time()
    os.makedirs(tmp, exist_ok=True)
    domains, seg_rows = [], []
    for pth, chain, chopping in zip(args.input, chains, choppings):
        t1 = time.time()
        doms = chop.domains_from_chopping(pth, chopping, chain)
        domains.extend(doms)
        seg_rows.append(chop.segment_row(pth, chopping, chain, runtime=time.time() - t1))
    if sharded.rank_world()[0] == 0:
        write_segment_results(results=seg_rows, output_file=args.output + "_segment.tsv", header=args.output_headers)
    if not domains:
        logging.info("easy-search finished after segmentation: no domains to search.")
        return
    _search_and_write(args, domains, True, None, fields, tmp)
    logging.info(f"Finished easy-search in {time.time() - t0:.3f} seconds.")
    shutil.rmtree(tmp, ignore_errors=True)


def createdb(argv) -> None:
    p = argparse.ArgumentParser(prog="createdb", description="Embed a directory of PDB files into a Foldclass database.",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("input_dir", type=str)
    p.add_argument("out_db", type=str)
    p.add_argument("-d", "--device", type=str, default="cuda")
    p.add_argument("--layout", type=str, default="pt", choices=["pt", "faiss", "both"])
    p.add_argument("--weights", type=str, default=None)
    args = p.parse_args(argv)
    _join_process_group(args)
    _log_command("createdb")
    t0 = time.time()
    run_createdb(pdb_files=args.input_dir, out_db=args.out_db, device=args.device, layout=args.layout, weights_path=args.weights)
    logging.info(f"Finished createdb in {time.time() - t0} seconds.")


def main(argv=None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    modes = {"search": search, "easy-search": easy_search, "createdb": createdb}
    if not argv or argv[0] not in modes:
        print("usage: python -m merizo_search_amd.cli {search,easy-search,createdb} ...  "
              "(segment: out of scope, use the reference)", file=sys.stderr)
        sys.exit(2)
    modes[argv[0]](argv[1:])
    sharded.finalize_distributed()        # (a rank that fails exits on its own; torchrun then ends the others)


if __name__ == "__main__":
    main()
