#!/usr/bin/env python
"""Foldclass search benchmark: queries/sec of the exact 128-d cosine top-k scan on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: starts its own ranks as children)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N = 1 (default): BASELINE.json configs[1] (C2) -- brute-force cosine top-10 over a 1M x 128 float32
synthetic database, batch = 256 queries.  A step is one pass of the hot path over one query batch:
raw query embeddings -> F.normalize INSIDE the calls (MS_MODE_IP_NORMQ, as the driver makes them) -> sample
pass -> fused Q.D^T + top-k scan of the resident shard -> merge of the per-stream lists [-> RCCL all-gather
of the per-shard top-k + shard merge when N > 1].  The database and the raw query embeddings are resident in HBM before the timed
region starts.

N > 1: the TED shape of BASELINE.json configs[3] (C4), weak scaling -- every rank generates and holds
45,625,000 rows (23.4 GB; 8 ranks = the 365M-row database), batch = 4096 queries, one all-gather
of 12*nq*k bytes per rank and one merge per step.  queries/s is then (nearly) constant in N while the
database grows N-fold; the N = 1 point of that curve is the `c4_shard` entry of the default run.
`--rows R` overrides either default with R total rows sharded over the ranks (strong scaling).

The top-level line is the fp32 scan (ms_ip_topk_prepare / _scan / _finish): the reference's own arithmetic, data-independent.  The
PREFILTERED search (ms_ip_topk_prefiltered: the rows scanned once with fp16 matrix instructions over the fp16 image built when the
database became resident -- MS_PF_F16X2: fp16 rows x split fp16 queries, 2 instructions per 16 dimensions, 256 B per row -- the best
2k rows per query re-scored with the exact fp32 chain, the answer proved complete per query, an exact fp32 pass for the queries whose
proof failed) -- what the driver runs for more than 64 queries, bit-identical results -- is the named block `prefiltered` beside it,
with its own dtype, roofs and traffic; `--no-prefilter` skips those blocks; MS_BENCH_PF_FORMAT=f16x1|bf16x3 times the other arithmetics.

ONE JSON line under 4 KB is printed on stdout by rank 0 (contract in the task statement: metric, value, ..., `roofline`, a short
`prefiltered` summary, `cpu_baseline`, headline numbers of the other entries under `more`); the FULL document -- every block below --
goes to bench_full.json beside this file and to stderr.  The blocks:
  roofline      dominant kernel of the top-level step, the fp32 scan launch (ms_scan_loader_kernel; ms_scan_kernel below 3 query
                tiles): algorithmic flops (2*128*nq*rows per launch) over the HIP-event duration of the launch against the fp32
                MFMA peak (157.3 TFLOP/s) when nq >= 39, else algorithmic bytes (512 B per row) against the 8 TB/s HBM peak; both
                fractions always included, plus `step_frac`: the same work over the whole step time (what the user gets);
                `traffic` (HBM bytes per launch) of the TOP-LEVEL kernel is measured by this run itself: after the timed regions two
                `rocprofv3 --kernel-trace --pmc` child passes (FETCH_SIZE; WRITE_SIZE) of this script on the same workload, gfx950
                corrections applied (`--no-live-traffic`, or no rocprofv3: the figure of the committed PMC passes under profiles/, and
                `traffic_from` says which); the other blocks carry the committed profiles' figures (`traffic_source`);
  prefiltered   (N = 1) the same step through the prefiltered search: ms per step, q/s, identical_to_fp32, how many queries needed
                the exact pass, and the roofs of ITS scan launch (ms_scan_pf16_kernel): the flops it executes (m 16-bit matrix
                instructions per 16 dimensions) against the dense 16-bit matrix peak, the bytes of the image (256 B per row)
                against the HBM peak;
  two_in_flight (N = 1, informative) the same steps with two query batches in flight on two HIP streams;
  streamed      (N = 1) the reference's db_iterator loop: a host memmap searched block by block over PCIe (bound: 63 GB/s);
  collective    (N > 1) backend, ranks, DISTINCT devices (PCI bus ids), RCCL version, the exchange's own time;
  clustered     (N = 1) C2 with 64 of the 256 queries owning a family of 200 near-duplicate rows (within 1e-6 of each other): the
                prefilter cannot prove those 64 answers and gives exactly them an exact pass; both paths timed;
  cpu_baseline  the CPU oracle (oracle/oracle.c: AVX2 + OpenMP port of the faiss path) on this host's
                cores, and under `torch_cpu` the reference's own torch op shapes (oracle/torch_baseline.py)
                -- per-query cosine_similarity*mask->topk, blockwise-262,144 normalize->Q@D^T->topk->merge,
                batch=1 EGNN loop; all on bounded samples;
  hbm_regime    (N = 1) nq = 1 / 4 / 8 / 32 over 1M, 4M and 45.6M rows: GB/s against the HBM peak;
  c4_shard      (N = 1) one rank's share of C4: 45,625,000 rows x 4096 queries, fp32 scan and prefiltered;
  k_sweep       (N = 1) the C2 shape at k = 1 / 10 / 20 / 32 / 64, both paths;
  c3_search     (N = 1) C3's search half: 500k rows of a `.pt` database (normalised once at load, as the product does),
                1000 queries, cosine + length mask, fp32 scan and prefiltered;
  embed         (N = 1) C3's embed half: 1000 TED-length domains -> embeds/s and fraction of the fp32 MFMA
                peak, and the C5 query (AF-Q96PD2, 3 domains) latency;
  c3_end_to_end (N = 1) C3 as ONE timed unit: embed 1000 domains, then search their embeddings against the 500k-row database.
`--no-extras` skips everything from hbm_regime down, `--no-cpu-baseline` the CPU legs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_F32_PEAK = 157.3e12     # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
HBM_PEAK = 8.0e12            # same guide, "HBM3E peak BW" (spec)
MFMA_BF16_PEAK = 2.5e15      # same guide, dense bf16 matrix peak
C4_ROWS_PER_GPU = 45_625_000
C4_NQ = 4096


def scan_kernel_name(nq, k):
    return "ms_scan_loader_kernel" if (nq > 64 and k <= 64) else "ms_scan_kernel"


def small_batch_note(nq, ops=None):
    """What one step of <= 64 queries launches, from the thresholds the loaded library applies (ms_small_batch_thresholds)."""
    if nq > 64:
        return None
    fused, inkernel = ops.small_batch_thresholds() if ops is not None else (2, 4)
    norm = "normalise in the scan's prologue" if nq <= inkernel else "query preparation launch (F.normalize)"
    if nq <= fused:
        what = "ONE launch (%s, merge by the scan's last workgroup)" % norm
    elif nq < 8:
        what = "%s%s scan + merge launches (no sample pass below 8 queries)" % (norm, "," if nq <= inkernel else " +")
    else:
        what = "%s%s sample pass + bound + scan + merge launches" % (norm, "," if nq <= inkernel else " +")
    return ("one ms_ip_topk call per step (MS_MODE_IP_NORMQ); ms_per_step = wall time over back-to-back calls without event markers; "
            "scan_ms = HIP events around that call (a separate loop) = " + what)


PF_FORMATS = {   # name -> (matrix instructions per 16 dimensions, image bytes per row, matrix instruction, dtype label)
    "f16x2": (2, 256.0, "v_mfma_f32_32x32x16_f16", "f16x2 image scan + f32 re-score"),        # fp16 rows x split fp16 queries
    "f16x1": (1, 256.0, "v_mfma_f32_32x32x16_f16", "f16x1 image scan + f32 re-score"),        # fp16 rows x fp16 queries
    "bf16x3": (3, 512.0, "v_mfma_f32_32x32x16_bf16", "bf16x3 image scan + f32 re-score"),     # bf16 hi + lo of rows and queries
}


def pf_format_name(ops, image):
    return {ops.PF_F16X2: "f16x2", ops.PF_F16X1: "f16x1", ops.PF_BF16X3: "bf16x3"}[image.format]


def roofline(nq, rows, k, scan_ms, step_ms, prefiltered=None):
    """Both roofs for one scan launch over `rows` rows; the binding one is `frac`.
    prefiltered (a PF_FORMATS name): the scan of ms_ip_topk_prefiltered over the image issues m 16-bit matrix instructions per 16
    dimensions (f16x2: rowh.qh + rowh.ql; f16x1: rowh.qh; bf16x3: hi.hi, hi.lo, lo.hi): its matrix roof is m x the algorithmic flops
    over the dense 16-bit matrix peak (2.5 PFLOP/s for fp16 and bf16 alike), its memory roof the one read of the image (256 B per
    row for the fp16 image, 512 B for the split-bf16 one)."""
    flops = 2.0 * 128 * nq * rows
    bytes_ = 512.0 * rows
    t = scan_ms * 1e-3
    mfma_frac, hbm_frac = flops / t / MFMA_F32_PEAK, bytes_ / t / HBM_PEAK
    if prefiltered:
        m, bpr, instr, _ = PF_FORMATS[prefiltered]
        bytes_ = bpr * rows
        hbm_frac = bytes_ / t / HBM_PEAK
        executed = float(m) * flops
        bf16_frac = executed / t / MFMA_BF16_PEAK
        matrix_bound = executed / MFMA_BF16_PEAK >= bytes_ / HBM_PEAK
        if matrix_bound:
            roof = {"bound": "mfma", "achieved": executed / t / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": bf16_frac,
                    "step_frac": executed / (step_ms * 1e-3) / MFMA_BF16_PEAK}
        else:
            roof = {"bound": "hbm", "achieved": bytes_ / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm_frac,
                    "step_frac": bytes_ / (step_ms * 1e-3) / HBM_PEAK}
        roof.update({"traffic": None, "kernel": "ms_scan_pf16_kernel (%s)" % prefiltered if prefiltered != "bf16x3" else "ms_scan_pf2_kernel (bf16x3)",
                     "kernel_ms": scan_ms, "hbm_frac": hbm_frac, "matrix_16bit_frac": bf16_frac, "format": prefiltered, "matrix_instruction": instr,
                     "matrix_instructions_per_16_dims": m, "image_bytes_per_row": bpr,
                     "algorithmic_tflops": flops / t / 1e12, "rows_per_launch": rows, "algorithmic_bytes_per_launch": bytes_,
                     "algorithmic_flops_per_launch": flops, "executed_16bit_flops_per_launch": executed,
                     "note": "see notes.prefiltered_roofline"})
        return roof
    if nq >= 39:
        roof = {"bound": "mfma", "achieved": flops / t / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": mfma_frac,
                "step_frac": flops / (step_ms * 1e-3) / MFMA_F32_PEAK}
    else:
        roof = {"bound": "hbm", "achieved": bytes_ / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm_frac,
                "step_frac": bytes_ / (step_ms * 1e-3) / HBM_PEAK}
    roof.update({"traffic": None, "kernel": scan_kernel_name(nq, k), "kernel_ms": scan_ms,
                 "kernel_ms_from": "HIP events around the scan launch of every %d-th step of the timed region (every step when there are fewer than %d)" % (EVENT_STRIDE, 4 * EVENT_STRIDE),
                 "mfma_frac": mfma_frac, "hbm_frac": hbm_frac,
                 "rows_per_launch": rows, "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bytes_})
    return roof


def attach_committed_traffic(roof, pmc_name):
    """HBM traffic of one launch of the entry's kernel, from the PMC passes of the same workload committed under profiles/
    (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 2x read correction: tools/pmc_to_json.py).  NOT measured in
    this run: counters need the profiler; the fields say so."""
    path = os.path.join(REPO, "profiles", pmc_name)
    if not os.path.exists(path):
        return
    with open(path) as fh:
        pmc = json.load(fh)
    roof["traffic"] = pmc.get("traffic_bytes_per_launch")
    roof["traffic_from_committed_profile"] = True
    roof["traffic_source"] = "profiles/%s" % pmc_name
    roof["traffic_from"] = "profiles/%s (committed rocprofv3 --pmc passes of this workload)" % pmc_name
    if pmc.get("matrix_pipe_busy_fraction") is not None:
        roof["matrix_pipe_busy_fraction_from_committed_profile"] = pmc["matrix_pipe_busy_fraction"]


def measure_traffic_live(kernels, log, timeout_s=150):
    """HBM bytes per launch of the top-level step's scan kernel, measured NOW: two `rocprofv3 --kernel-trace --pmc <counter>` CHILD passes
    (FETCH_SIZE; WRITE_SIZE: counters that cannot share a pass) of this very script on the same workload (`--no-extras --no-cpu-baseline
    --no-live-traffic --steps 5 --warmup 3`), corrected as MI355X_MICROARCH.md prescribes (KiB units; gfx950 reports half of a wide /
    LDS-DMA read stream: reads = 2 x FETCH_SIZE x 1024).  The children are started as fresh processes (nothing is exec'd in this one) under
    a hard time limit and killed as a group past it.  -> (bytes per launch, launches counted) or None (no rocprofv3, a pass failed or
    timed out: the caller keeps the figure of the committed profile and says so).  `kernels`: {key: kernel-name substring}; -> {key:
    (bytes per launch, launches counted)} for the kernels both passes saw."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None
    means = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ms_pmc_")
        cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", os.path.realpath(sys.executable), os.path.abspath(__file__),
               "--no-extras", "--no-cpu-baseline", "--no-live-traffic", "--no-pipelined", "--steps", "5", "--warmup", "3"]
        env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["TMPDIR"] = tempfile.gettempdir()
        proc = subprocess.Popen(cmd, cwd=tempfile.gettempdir(), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()
            rc = -9
        vals = {key: [] for key in kernels}
        if rc == 0:
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] != ctr:
                            continue
                        for key, name in kernels.items():
                            if name in r["Kernel_Name"]:
                                vals[key].append(float(r["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not any(vals.values()):
            log("live traffic: the %s pass gave nothing (rc %s): keeping the committed profiles' figures" % (ctr, rc))
            return None
        means[ctr] = {key: (float(np.mean(v)), len(v)) for key, v in vals.items() if v}
    out = {}
    for key in kernels:
        if key in means["FETCH_SIZE"] and key in means["WRITE_SIZE"]:
            out[key] = (2.0 * means["FETCH_SIZE"][key][0] * 1024.0 + means["WRITE_SIZE"][key][0] * 1024.0, means["FETCH_SIZE"][key][1])
    return out or None


class SearchBench:
    """One shard resident on this rank + a query batch; `step()` is the timed unit."""

    def __init__(self, torch, dist, ops, syn, sharded, dev, rank, world, n_total, lo, hi, nq, k, exchange=False, prefilter=False,
                 clustered=0):
        self.torch, self.dist, self.ops, self.world, self.sharded = torch, dist, ops, world, sharded
        self.n_total, self.lo, self.n_local, self.nq, self.k = n_total, lo, hi - lo, nq, k
        self.exchange = exchange or world > 1
        # synthetic inputs: unit rows ~ N(0,1)/|.| (seed 0, independent of the sharding); raw
        # queries ~ N(0,1) (seed 1); 3 near-duplicates of every query planted at known global rows
        self.db = syn.device_database(self.n_local, lo, seed=0, device=dev, normalize=True)
        g = torch.Generator(device=dev); g.manual_seed(1)
        self.q_raw = torch.randn((nq, 128), generator=g, device=dev, dtype=torch.float32) * 3.0
        qn = (self.q_raw / self.q_raw.norm(dim=1, keepdim=True)).cpu()
        gp = torch.Generator(device="cpu"); gp.manual_seed(2)
        # nq * 3 distinct pseudo-random global rows without materialising a permutation of n_total (365M at C4)
        step_ = 2_147_483_629 if n_total % 2_147_483_629 else 2_147_483_587                    # primes: a bijection modulo n_total
        self.planted = ((torch.arange(nq * 3, dtype=torch.int64) * step_ + 12_345) % n_total).reshape(nq, 3)
        near = qn[:, None, :] + torch.randn((nq, 3, 128), generator=gp) * 0.02
        near = near / near.norm(dim=2, keepdim=True)
        flat = self.planted.reshape(-1)
        mine = (flat >= lo) & (flat < hi)
        if mine.any():
            self.db[(flat[mine] - lo).to(dev)] = near.reshape(-1, 128)[mine].to(dev)
        self.clustered = int(clustered)
        if self.clustered:
            # `clustered` of the queries own a family of 200 rows within ~1e-6 of each other (copies of the query's direction with
            # 2e-7 noise: the construction of tools/stress_prefilter.py): no prefilter can prove those answers
            owners = torch.arange(0, nq, max(1, nq // self.clustered))[: self.clustered]
            fam = qn[owners][:, None, :] + torch.randn((len(owners), 200, 128), generator=gp) * 2e-7
            fam = fam / fam.norm(dim=2, keepdim=True)
            rows = ((torch.arange(len(owners) * 200, dtype=torch.int64) * step_ + 777_777) % n_total)
            mine = (rows >= lo) & (rows < hi)
            self.db[(rows[mine] - lo).to(dev)] = fam.reshape(-1, 128)[mine].to(dev)
            self.cluster_owners = owners
        self.q = torch.empty_like(self.q_raw)
        self.ex = sharded.PackedExchange(nq, k, dev)     # this rank's results are written straight into its all-gather block
        self.image = self.row_norm_bound = None
        self._set_path(prefilter)

    def _set_path(self, prefilter):
        """fp32 scan (ms_ip_topk stages) or, for the shapes it serves, the prefiltered search as the driver runs it on a resident
        shard: split image built once, the largest row norm measured once."""
        ops, dev = self.ops, self.db.device
        # (decided on the smallest shard so that every rank takes the same path: the steps contain collectives)
        self.prefilter = bool(prefilter) and (ops.prefilter_serves(self.n_total // self.world, self.nq, self.k) or
                                              ops.prefilter_serves(self.n_total // self.world, self.nq, self.k, ops.pf_default_format()))
        if self.prefilter and self.image is None:
            self.row_norm_bound = float(1.0 / ops.row_inv_norms(self.db, 1e-30).min()) * (1.0 + 1e-6)
            self.image = ops.pf_build_image(self.db, row_norm_bound=self.row_norm_bound)     # (as the driver does: the fp16 image, then
            if ops.pf_format_is_auto():                                                       #  F16X1 or F16X2 by searching 256 of its rows)
                self.image = ops.pf_choose_format(self.db, self.image, self.row_norm_bound)
            fmt_env = os.environ.get("MS_BENCH_PF_FORMAT")                                    # A/B runs: f16x1 over the same image
            if fmt_env:
                self.image = self.image.as_format({"f16x2": ops.PF_F16X2, "f16x1": ops.PF_F16X1, "bf16x3": ops.PF_BF16X3}[fmt_env])
        if self.prefilter and self.nq <= 32 and self.image.format == ops.PF_F16X1:
            self.image = self.image.as_format(ops.PF_F16X2)       # (as the engine does for a few queries: the tighter arithmetic at the same HBM-bound speed)
        self.ws = self.torch.empty_like((ops.PrefilterWorkspace if self.prefilter else ops.TopKWorkspace)(dev).get(self.n_local, self.nq, self.k))

    def variant(self, prefilter):
        """The same database and queries through the other path."""
        b = SearchBench.__new__(SearchBench)
        b.__dict__.update(self.__dict__)
        b.ex = self.sharded.PackedExchange(self.nq, self.k, self.db.device)
        b._set_path(prefilter)
        return b

    def step(self, events=None):
        ops, ex = self.ops, self.ex
        if self.nq <= 64:
            # the reference's own CLI regime (a few query domains, dbsearch.py:531-546): ONE C-ABI call, as the driver makes it
            # (dbsearch.knn_exact(raw_queries=True)) -- F.normalize and, for a handful of queries, the merge inside the scan launch
            if events is not None:
                events[0].record()
            if self.prefilter:      # (>= ms_pf_few_min_rows() rows: the same call over the fp16 image -- half the bytes of the fp32 rows)
                ops.ip_topk_prefiltered(self.db, self.q_raw, self.k, self.row_norm_bound, mode=ops.MODE_IP_NORMQ, row_offset=self.lo, workspace=self.ws,
                                        out=(ex.out_s, ex.out_i), image=self.image)
            else:
                ops.ip_topk(self.db, self.q_raw, self.k, mode=ops.MODE_IP_NORMQ, row_offset=self.lo, workspace=self.ws, out=(ex.out_s, ex.out_i))
            if events is not None:
                events[1].record()
            if self.exchange:
                ex.exchange()
                return ex.merge()
            return ex.out_s, ex.out_i
        # F.normalize of the batch's raw embeddings (dbsearch.py:303-304) is INSIDE the calls (MS_MODE_IP_NORMQ, as the driver makes them:
        # dbsearch.knn_exact(raw_queries=True)): the fp32 path's preparation launch normalises and pads; the prefiltered path over the fp16
        # image has no preparation launch at all (round 6: the sample pass and the scan normalise in their set-up, the re-scoring exactly)
        if self.prefilter:
            kw = dict(row_norm_bound=self.row_norm_bound, image=self.image, mode=ops.MODE_IP_NORMQ)
            ops.ip_topk_prefiltered_stage("prepare", self.db, self.q_raw, self.k, self.ws, **kw)   # sample pass on approximate scores
            if events is not None:
                events[0].record()
            ops.ip_topk_prefiltered_stage("scan", self.db, self.q_raw, self.k, self.ws, **kw)      # dominant kernel: ONE scan launch
            if events is not None:
                events[1].record()
            # merge of the candidate lists, exact re-scoring + proof, and the exact pass over the flagged queries (normally none)
            ops.ip_topk_prefiltered_stage("finish", self.db, self.q_raw, self.k, self.ws, out=(ex.out_s, ex.out_i), row_offset=self.lo, **kw)
            if self.exchange:
                ex.exchange()
                return ex.merge()
            return ex.out_s, ex.out_i
        ops.ip_topk_prepare(self.db, self.q_raw, self.k, self.ws, mode=ops.MODE_IP_NORMQ)      # F.normalize + sample pass (lower bound per query)
        if events is not None:
            events[0].record()
        ops.ip_topk_scan(self.db, self.q_raw, self.k, self.ws, mode=ops.MODE_IP_NORMQ)         # dominant kernel: ONE scan launch
        if events is not None:
            events[1].record()
        ops.ip_topk_finish(self.n_local, self.nq, self.k, self.ws, ex.out_s, ex.out_i, row_offset=self.lo)
        if self.exchange:
            ex.exchange()                                                   # ONE RCCL all-gather of 12*nq*k bytes per rank
            return ex.merge()                                               # merge of the S blocks in place
        return ex.out_s, ex.out_i

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup, prep_budget_s=0.3, max_prep=40):
        """-> (seconds for `steps` steps [max over ranks], mean scan-launch ms [max over ranks], last result)."""
        torch = self.torch
        self.step(); self.fence()
        t = time.perf_counter(); self.step(); self.fence(); est = time.perf_counter() - t
        for _ in range(min(max_prep, int(prep_budget_s / max(est, 1e-4)))):       # untimed preparation, like the data generation:
            self.step()                                                     # the GPU leaves its idle clocks within ~20 launches
        for _ in range(warmup):
            res = self.step()
        self.fence()
        # HIP events around the scan launch of every EVENT_STRIDE-th step of the timed region (a pair of markers costs the stream
        # ~3 us; around every launch they were 1 % of a C2 step): the launch duration is their mean
        stride = EVENT_STRIDE if steps >= 4 * EVENT_STRIDE else 1
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(0, steps, stride)]
        t0 = time.perf_counter()
        for s in range(steps):
            res = self.step(evs[s // stride] if s % stride == 0 else None)
        self.fence()
        elapsed = time.perf_counter() - t0
        scan_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        if self.world > 1:
            t = torch.tensor([elapsed, scan_ms], dtype=torch.float64, device=self.db.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed, scan_ms = float(t[0]), float(t[1])
        return elapsed, scan_ms, res

    def local_rate(self, steps):
        """queries/s of the ranks' own shard scans WITHOUT the exchange and the shard merge (slowest rank): what one GPU does on
        its share of the database -- the one-rank reference point of the weak-scaling curve, measured by the same processes."""
        keep, self.exchange = self.exchange, False
        try:
            self.step(); self.fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.fence()
            el = time.perf_counter() - t0
        finally:
            self.exchange = keep
        if self.world > 1:
            t = self.torch.tensor([el], dtype=self.torch.float64, device=self.db.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t[0])
        return self.nq * steps / el

    def check(self, res, sharded):
        """Correctness of what was timed, after the timed region: recall@k against an exact brute force
        (torch matmul + topk over every shard, merged over the ranks) -- for every query when the score
        matrix is affordable, else for the first 64 -- and the planted neighbours."""
        torch, k = self.torch, self.k
        fs, fi = res
        fi_c = fi.cpu()
        planted = float(np.mean([len(set(self.planted[j].tolist()) & set(fi_c[j].tolist())) / 3.0 for j in range(self.nq)]))
        nb = self.nq if self.n_local * self.nq <= (1 << 34) else min(self.nq, 64)
        kk = min(k + 4, max(self.n_local, 1))
        qn = self.q_raw[:nb] / self.q_raw[:nb].norm(dim=1, keepdim=True)
        bs = bi = None
        # (row chunks of at most 2^20: with 2^31 / nb rows per chunk -- a [64, 2^25] score matrix, 2^31 elements, in one matmul + topk --
        #  torch's brute force itself came out wrong at the C4 shard (recall 0.92 "against" it, 48 of 64 lists; 32-bit indexing or the
        #  library GEMM at that size: not investigated), while the same product in 2^19..2^20-row chunks agrees with the scan to 2e-6 for
        #  all 4096 queries (tests/test_fullsize_gpu.py) and gives 64 of 64 identical lists here)
        chunk = max(1, min(self.n_local, 1 << 20, (1 << 31) // max(nb, 1)))
        for r0 in range(0, self.n_local, chunk):
            top = torch.topk(qn @ self.db[r0:r0 + chunk].T, min(kk, self.n_local - r0), dim=1)
            idx = top.indices + (r0 + self.lo)
            if bs is None:
                bs, bi = top.values, idx
            else:
                cs, ci = torch.cat([bs, top.values], 1), torch.cat([bi, idx], 1)
                keep = torch.topk(cs, min(kk, cs.shape[1]), dim=1)
                bs, bi = keep.values, torch.gather(ci, 1, keep.indices)
        if self.world > 1:
            gs, gi = sharded.allgather_results(bs.contiguous(), bi.contiguous())
            bs, bi = self.ops.topk_merge(gs, gi)
        bs, bi, got = bs.cpu().numpy(), bi.cpu().numpy(), fi_c[:nb].numpy()
        kth = bs[:, min(k, bs.shape[1]) - 1]
        recall, identical = [], 0
        for j in range(nb):
            ok = set(bi[j][bs[j] >= kth[j] - 1e-6].tolist())               # rows tied with the k-th within 1e-6 are interchangeable
            recall.append(len(ok & set(got[j].tolist())) / float(min(k, bs.shape[1])))
            identical += int(np.array_equal(got[j], bi[j, :k]))
        return {"recall_at_k": float(np.mean(recall)), "recall_queries_checked": nb,
                "topk_identical_to_torch_bruteforce": "%d of %d queries" % (identical, nb), "planted_recall": planted}


EVENT_STRIDE = 4       # SearchBench.run: HIP events around every 4th scan launch of the timed region


def hbm_regime(make, rows_list, log):
    """nq = 1 / 4 / 8 / 32 (few queries / one query tile: the reference's own CLI usage, dbsearch.py:531-546) -- GB/s of the scan
    launch and of the whole step against the 8 TB/s HBM peak."""
    out = []
    for rows in rows_list:
        for nq in (1, 4, 8, 32):
            b = make(rows, nq)
            steps = 40 if rows <= 4_000_000 else 6
            _, scan_ms, _ = b.run(steps, 3, prep_budget_s=0.05)
            # the step time is taken over back-to-back calls WITHOUT event markers between them (two markers per call cost ~10 us of a
            # 120 us search: the run above, which has them, gives the launch duration only)
            reps = 100 if rows <= 4_000_000 else 10
            b.fence(); t0 = time.perf_counter()
            for _ in range(reps):
                b.step()
            b.fence()
            ms = (time.perf_counter() - t0) / reps * 1e3
            out.append({"rows": rows, "nq": nq, "k": b.k, "ms_per_step": ms, "scan_ms": scan_ms, "queries_per_s": nq / ms * 1e3,
                        "scan_GBps": 512.0 * rows / scan_ms / 1e6, "scan_frac_of_hbm_peak": 512.0 * rows / (scan_ms * 1e-3) / HBM_PEAK,
                        "step_frac_of_hbm_peak": 512.0 * rows / (ms * 1e-3) / HBM_PEAK, "kernel": scan_kernel_name(nq, b.k),
                        "note": "see notes.small_batch.nq%d" % nq})
            log("hbm_regime rows=%d nq=%d: call %.3f ms (%.1f%% of 8 TB/s), step %.3f ms (%.1f%%)" % (rows, nq, scan_ms, out[-1]["scan_frac_of_hbm_peak"] * 100, ms, out[-1]["step_frac_of_hbm_peak"] * 100))
            few_image_entry(b, out[-1], reps, log)
            del b
    return out


def few_image_entry(b, entry, reps, log):
    """<= 64 queries over the fp16 image (ms_ip_topk_prefiltered serves them from ms_pf_few_min_rows() rows on): the same step reading 256 B
    per row instead of 512 -- added to an hbm_regime entry as `image`."""
    res = b.step(); b.fence()
    bp = b.variant(True)
    if not bp.prefilter:
        return
    for _ in range(3):
        rp = bp.step()
    bp.fence(); t0 = time.perf_counter()
    for _ in range(reps):
        rp = bp.step()
    bp.fence()
    ms = (time.perf_counter() - t0) / reps * 1e3
    tq = b.torch
    entry["image"] = {"format": pf_format_name(bp.ops, bp.image), "ms_per_step": ms, "queries_per_s": b.nq / ms * 1e3,
                      "image_GBps": 256.0 * b.n_local / ms / 1e6, "image_frac_of_hbm_peak": 256.0 * b.n_local / (ms * 1e-3) / HBM_PEAK,
                      "fp32_rows_equivalent_frac_of_hbm_peak": 512.0 * b.n_local / (ms * 1e-3) / HBM_PEAK,
                      "identical_to_fp32": bool(tq.equal(rp[1], res[1]) and tq.equal(rp[0].view(tq.int32), res[0].view(tq.int32))),
                      "exact_pass_queries": bp.ops.prefilter_flagged(bp.ws)}
    log("   over the fp16 image (%s): step %.3f ms = %.1f%% of 8 TB/s reading 256 B per row, identical: %s" % (
        entry["image"]["format"], ms, entry["image"]["image_frac_of_hbm_peak"] * 100, entry["image"]["identical_to_fp32"]))
    del bp


def _timed_stages(torch, step, steps, warm=10):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t = time.perf_counter()
    for s_ in range(steps):
        step(evs[s_])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3, float(np.mean([a.elapsed_time(b) for a, b in evs]))


def c3_search_bench(torch, ops, syn, dev, k, log, prefilter=True):
    """C3's search half: a `.pt`-style database of 500,000 RAW rows, 1000 query embeddings, cosine + length mask
    (search_query_against_db, dbsearch.py:75-81; mincov 0.7).  As in the product (foldclass/engine.py:cosine_rows) the rows are
    L2-normalised once when the database becomes resident and every search runs in MS_MODE_COSINE_UNIT: through the fp32 scan
    (top-level numbers of the entry) and through the prefiltered search over the split image of the unit rows (`prefiltered`)."""
    n, nq, mincov = 500_000, 1000, 0.7
    db = syn.device_database(n, 0, seed=3, device=dev, normalize=False) * 2.5
    lengths = torch.from_numpy(syn.ted_lengths(n, seed=4).astype(np.float32)).to(dev)
    qlen = torch.from_numpy(syn.ted_lengths(nq, seed=5).astype(np.float32)).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(6)
    q = torch.randn((nq, 128), generator=g, device=dev, dtype=torch.float32)
    unit = ops.l2_normalize_rows_(db.clone(), 1e-8)          # what the engine keeps resident
    ws = torch.empty_like(ops.TopKWorkspace(dev).get(n, nq, k))
    out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=lengths, qlen=qlen, mincov=mincov)

    def step(ev=None):
        ops.ip_topk_prepare(unit, q, k, ws, **kw)
        if ev is not None:
            ev[0].record()
        ops.ip_topk_scan(unit, q, k, ws, **kw)
        if ev is not None:
            ev[1].record()
        ops.ip_topk_finish(n, nq, k, ws, out_s, out_i)

    ms, scan_ms = _timed_stages(torch, step, 40)
    # check: masked cosine of the returned rows, recomputed in float64, and order
    rows = db[out_i.reshape(-1)].double().reshape(nq, k, 128)
    cos = (rows * q.double()[:, None, :]).sum(2) / rows.norm(dim=2) / q.double().norm(dim=1)[:, None]
    cos = cos * (qlen[:, None] >= lengths[out_i] * mincov).double()
    err = float((cos - out_s.double()).abs().max())
    out = {"workload": "C3 search half: 500,000 x 128 RAW fp32 rows (.pt layout), 1000 queries, cosine + length mask (mincov 0.7), top-%d" % k,
           "ms_per_step": ms, "queries_per_s": nq / ms * 1e3, "max_abs_score_error_vs_float64": err,
           "roofline": roofline(nq, n, k, scan_ms, ms)}
    out["roofline"]["kernel"] = "ms_scan_loader_kernel<5, 2> (unit-row cosine variant: in-chain filter on the final scores, length mask in the rare path)"
    out["roofline"]["algorithmic_bytes_per_launch"] = 516.0 * n          # rows + their lengths
    attach_committed_traffic(out["roofline"], "r06_c3_pmc.json")
    log("c3_search: %.3f ms per 1000-query batch (scan %.3f ms = %.1f%% of fp32 MFMA peak), score error %.1e" % (ms, scan_ms, out["roofline"]["frac"] * 100, err))
    state = {"unit": unit, "lengths": lengths, "mincov": mincov, "n": n, "k": k, "image": None, "pws": None}
    if prefilter and ops.prefilter_serves(n, nq, k):
        img = ops.pf_build_image(unit, row_norm_bound=1.0 + 1e-5)
        if ops.pf_format_is_auto():
            img = ops.pf_choose_format(unit, img, 1.0 + 1e-5)
        pws = torch.empty_like(ops.PrefilterWorkspace(dev).get(n, nq, k))
        ps, pi = torch.empty_like(out_s), torch.empty_like(out_i)
        pkw = dict(row_norm_bound=1.0 + 1e-5, image=img, **kw)

        def pstep(ev=None):
            ops.ip_topk_prefiltered_stage("prepare", unit, q, k, pws, **pkw)
            if ev is not None:
                ev[0].record()
            ops.ip_topk_prefiltered_stage("scan", unit, q, k, pws, **pkw)
            if ev is not None:
                ev[1].record()
            ops.ip_topk_prefiltered_stage("finish", unit, q, k, pws, out=(ps, pi), **pkw)

        pms, pscan = _timed_stages(torch, pstep, 40)
        out["prefiltered"] = {"ms_per_step": pms, "queries_per_s": nq / pms * 1e3, "dtype": PF_FORMATS[pf_format_name(ops, img)][3],
                              "identical_to_fp32": bool(torch.equal(pi, out_i) and torch.equal(ps.view(torch.int32), out_s.view(torch.int32))),
                              "exact_pass_queries": ops.prefilter_flagged(pws), "roofline": roofline(nq, n, k, pscan, pms, prefiltered=pf_format_name(ops, img))}
        attach_committed_traffic(out["prefiltered"]["roofline"], "r06_pf_c3_pmc.json")
        log("c3_search prefiltered: %.3f ms per batch (scan %.3f ms), identical: %s, exact-pass queries: %d" % (
            pms, pscan, out["prefiltered"]["identical_to_fp32"], out["prefiltered"]["exact_pass_queries"]))
        state.update(image=img, pws=pws)
    del db, ws
    return out, state


def c3_end_to_end(torch, ops, enc, coords, lens, state, log):
    """C3 as BASELINE.json states it -- embed 1,000 domains, then search them against the 500k-row database -- as ONE timed unit:
    CA coordinates on the host -> ragged encoder launches -> embeddings stay on the device -> cosine + length-mask top-k (the
    prefiltered search when the image exists, as the driver would run it) -> results on the device."""
    dev = state["unit"].device
    qlen = torch.from_numpy(np.asarray(lens, dtype=np.float32)).to(dev)
    k, n = state["k"], state["n"]
    nq = len(coords)
    ws = state["pws"] if state["image"] is not None else torch.empty_like(ops.TopKWorkspace(dev).get(n, nq, k))
    kw = dict(mode=ops.MODE_COSINE_UNIT, lengths=state["lengths"], qlen=qlen, mincov=state["mincov"])

    def unit_of_work():
        emb = enc.embed(coords)
        if state["image"] is not None:
            return ops.ip_topk_prefiltered(state["unit"], emb, k, 1.0 + 1e-5, workspace=ws, image=state["image"], **kw)
        return ops.ip_topk(state["unit"], emb, k, workspace=ws, **kw)

    unit_of_work(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); unit_of_work(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    sec = float(np.median(ts))
    log("c3_end_to_end: %.1f ms for embed 1000 + search 500k = %.0f domains/s" % (sec * 1e3, nq / sec))
    return {"workload": "C3: Foldclass EGNN embed %d TED-length domains + cosine/length-mask top-%d search of their embeddings over a 500,000 x 128 database, one timed unit" % (nq, k),
            "seconds": sec, "domains_per_s": nq / sec, "search_path": "prefiltered (split image)" if state["image"] is not None else "fp32 scan"}


def embed_bench(torch, ops, log):
    """C3's embed half (1000 TED-length domains in ragged launches) and the C5 query."""
    from merizo_search_amd.foldclass import synthetic as syn, weights as W
    from merizo_search_amd.foldclass.chopping import domains_from_chopping
    sd = W.synthetic_state_dict(0)
    weights, pe = W.pack_state_dict(sd)
    enc = ops.EgnnEncoder(weights, pe, "cuda:0")
    lens = syn.ted_lengths(1000, seed=5)
    coords = [syn.random_walk(int(n), seed=9000 + i) for i, n in enumerate(lens)]
    flops = float(sum(2.0 * (263680.0 * n * n + 525312.0 * n) for n in lens.astype(np.float64)))    # SURVEY.md 8d (restructured minimum)

    def timed(batch, reps):
        enc.embed(batch); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); enc.embed(batch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        return float(np.median(ts))

    t1000 = timed(coords, 3)
    split = os.environ.get("MS_EGNN_SPLIT", "1") != "0"
    # the edge GEMM (97 % of the flops) runs on split-bf16 matrix instructions by default: 6 bf16 instructions per 8 fp32 ones' worth
    # of k -> its matrix roof is flops * 6 / 2.5 PFLOP/s (= 416.7 TFLOP/s of algorithmic flops); MS_EGNN_SPLIT=0: fp32 instructions
    peak_alg = MFMA_BF16_PEAK / 6.0 if split else MFMA_F32_PEAK
    out = {"workload": "1000 synthetic domains, TED length distribution (sum N^2 = %.3g), 2-layer EGNN, ragged launches" % float((lens.astype(np.float64) ** 2).sum()),
           "seconds": t1000, "embeds_per_s": 1000.0 / t1000, "algorithmic_tflops": flops / t1000 / 1e12,
           "edge_gemm": "split-bf16 (3 x 3-way split operands, 6 v_mfma_f32_32x32x16_bf16 per 16 k; fp32-grade results)" if split else "fp32 (v_mfma_f32_32x32x2_f32)",
           "frac_of_fp32_mfma_peak": flops / t1000 / MFMA_F32_PEAK,
           "roofline": {"bound": "mfma", "achieved": flops / t1000 / 1e12, "peak": peak_alg / 1e12, "unit": "TFLOP/s",
                        "frac": flops / t1000 / peak_alg, "kernel": "ms_egnn_edge_kernel<%s> (+ proj / node / pool: whole encoder timed)" % ("true" if split else "false"),
                        "algorithmic_flops": flops,
                        "note": ("peak = the dense bf16 matrix peak / 6: the edge GEMM executes 6 bf16 matrix instructions where 8 fp32 ones of a sixteenth "
                                 "the rate would do (operands split three ways: hi, mid, lo); achieved counts ALGORITHMIC flops") if split else
                                "peak = the fp32 matrix peak"}}
    pdb = os.path.join(REPO, "tests", "golden", "AF-Q96PD2-F1-model_v4_ca.pdb")
    if os.path.exists(pdb):
        doms = domains_from_chopping(pdb, "71-189,190-290,291-453", "A")
        t3 = timed([d["coords"] for d in doms], 20)
        out["c5_query"] = {"structure": "AF-Q96PD2 (775 residues), chopping 71-189,190-290,291-453", "domain_lengths": [len(d["seq"]) for d in doms],
                           "embed_ms_three_domains": t3 * 1e3}
        whole = np.concatenate([d["coords"] for d in domains_from_chopping(pdb, "1-775", "A")])
        tw = timed([whole], 10)
        fw = 2.0 * (263680.0 * len(whole) ** 2 + 525312.0 * len(whole))
        out["c5_query"]["embed_ms_whole_chain_N%d" % len(whole)] = tw * 1e3
        out["c5_query"]["whole_chain_frac_of_roof"] = fw / tw / peak_alg
    log("embed: %.1f ms per 1000 domains = %.0f embeds/s = %.1f TFLOP/s algorithmic = %.1f%% of the %s roof" % (
        t1000 * 1e3, out["embeds_per_s"], flops / t1000 / 1e12, out["roofline"]["frac"] * 100, "split-bf16 matrix" if split else "fp32 matrix"))
    return out, sd, coords, enc, lens


PCIE_PEAK = 63.0e9          # MI355X_MICROARCH.md: PCIe Gen5 x16 host link, one direction


def streamed_bench(torch, ops, syn, dev, k, log, sizes=(8_000_000, C4_ROWS_PER_GPU), nqs=(1, 256, 4096), block_rows=262_144):
    """The reference's OWN scale mechanism, timed (VERDICT r05 missing #4): a database that is not resident -- a host memmap of `dbfname_IP`
    -- searched block by block, `knn_exact(xq, db_iterator(memmap, 262144), k)` (dbsearch.py:233-243, dbutil.py:28-35; --search_batchsize,
    merizo.py:145): every block crosses PCIe (pinned double-buffered upload on a side stream, foldclass/engine.py:device_blocks) while the
    previous one is scanned, the per-block top-k lists are merged as they come (ResultHeap).  It is the path of every database larger
    than the HBM budget.  Bound: the host link, 63 GB/s.  Per (rows, nq): rows/s and H2D GB/s of the whole search, the same blocks copied
    with nothing consuming them (`copy_only`), the same blocks scanned from HBM (`scan_only`), and how much of the shorter leg the
    overlap hides; a sweep of the block size; the batch size at which scan and upload would balance."""
    import shutil
    import tempfile
    from merizo_search_amd.foldclass import dbsearch as ds, dbutil
    from merizo_search_amd.foldclass.engine import HipEngine
    import logging
    quiet = logging.getLogger("bench.streamed"); quiet.setLevel(logging.WARNING)
    engine = HipEngine(str(dev))
    out = {"workload": "knn_exact over a host memmap in blocks of %d rows (the reference's db_iterator loop), k = %d" % (block_rows, k),
           "bound": "pcie", "peak_GBps": PCIE_PEAK / 1e9, "entries": [], "block_sweep": []}
    try:
        avail_kb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable"))
    except Exception:
        avail_kb = 0
    cand = [d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d)]
    for n in sizes:
        need = n * 512
        if avail_kb * 1024 < 3 * need + (16 << 30):
            out["entries"].append({"rows": n, "skipped": "host memory: %.0f GB available" % (avail_kb / 1e6)})
            continue
        where = next((d for d in cand if shutil.disk_usage(d).free > need + (4 << 30)), None)
        if where is None:
            out["entries"].append({"rows": n, "skipped": "no room for a %.1f GB file" % (need / 1e9)})
            continue
        path = os.path.join(where, "ms_bench_stream_%d_%d.db" % (os.getpid(), n))
        try:
            w = np.memmap(path, dtype=np.float32, mode="w+", shape=(n, 128))
            step = 1 << 21
            for r0 in range(0, n, step):                      # unit rows, the generator of every other block (row r is the same row anywhere)
                r1 = min(n, r0 + step)
                w[r0:r1] = syn.device_database(r1 - r0, r0, seed=0, device=dev, normalize=True).cpu().numpy()
            w.flush(); del w
            db = dbutil.db_memmap(path, (n, 128))
            g = torch.Generator(device=dev); g.manual_seed(1)
            q_all = torch.randn((max(nqs), 128), generator=g, device=dev, dtype=torch.float32) * 3.0

            def copy_only(rows_per_block):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for blk in engine.device_blocks(b for b in dbutil.db_iterator(db, rows_per_block)):
                    pass
                torch.cuda.synchronize()
                return time.perf_counter() - t0

            def stream(q, rows_per_block):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res = ds.knn_exact(q, dbutil.db_iterator(db, rows_per_block), k, engine, log=quiet, raw_queries=True, to_host=False)
                torch.cuda.synchronize()
                return time.perf_counter() - t0, res

            copy_only(block_rows)                             # (pins the staging buffers, warms the page cache)
            t_copy = min(copy_only(block_rows) for _ in range(2))
            resident = engine.upload_rows(db, 0, n) if torch.cuda.mem_get_info(dev)[0] > need + (8 << 30) else None
            for nq in nqs:
                q = q_all[:nq].contiguous()
                stream(q, block_rows)
                t_s, (ss, si) = min((stream(q, block_rows) for _ in range(2)), key=lambda x: x[0])
                ent = {"rows": n, "nq": nq, "block_rows": block_rows, "seconds": t_s, "rows_per_s": n / t_s, "h2d_GBps": need / t_s / 1e9,
                       "frac_of_pcie_peak": need / t_s / PCIE_PEAK, "queries_per_s": nq / t_s, "copy_only_seconds": t_copy,
                       "copy_only_GBps": need / t_copy / 1e9}
                if resident is not None:
                    blocks = [resident[r0:r0 + block_rows] for r0 in range(0, n, block_rows)]
                    ds.knn_exact(q, blocks, k, engine, log=quiet, raw_queries=True, to_host=False); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rs, ri = ds.knn_exact(q, blocks, k, engine, log=quiet, raw_queries=True, to_host=False)
                    torch.cuda.synchronize()
                    t_scan = time.perf_counter() - t0
                    ent["scan_only_seconds"] = t_scan
                    ent["identical_to_resident_blocks"] = bool(torch.equal(ri, si) and torch.equal(rs.view(torch.int32), ss.view(torch.int32)))
                    # how much of the SHORTER leg ran behind the longer one: 1 = the search takes max(copy, scan), 0 = their sum
                    ent["overlap_hidden_frac"] = max(0.0, min(1.0, (t_copy + t_scan - t_s) / max(min(t_copy, t_scan), 1e-9)))
                    # queries per batch at which the per-row scan time would equal the per-row upload time (fp32 scan: linear in nq above 64)
                    if nq > 64:
                        ent["balance_nq_estimate"] = int(nq * t_copy / max(t_scan, 1e-9))
                out["entries"].append(ent)
                log("streamed rows=%d nq=%d: %.3f s = %.1f GB/s over PCIe (%.0f%% of 63 GB/s; copy alone %.1f GB/s%s)" % (
                    n, nq, t_s, ent["h2d_GBps"], ent["frac_of_pcie_peak"] * 100, ent["copy_only_GBps"],
                    "; scan alone %.3f s, %.0f%% of the shorter leg hidden" % (ent["scan_only_seconds"], ent["overlap_hidden_frac"] * 100) if resident is not None else ""))
            if n == sizes[0]:
                for rb in (65_536, 262_144, 1_048_576):
                    q = q_all[:256].contiguous()
                    stream(q, rb)
                    t_b = min(stream(q, rb)[0] for _ in range(2))
                    out["block_sweep"].append({"rows": n, "nq": 256, "block_rows": rb, "seconds": t_b, "h2d_GBps": need / t_b / 1e9})
            del resident, db
        except (OSError, MemoryError, RuntimeError) as exc:
            out["entries"].append({"rows": n, "skipped": "%s: %s" % (type(exc).__name__, str(exc)[:120])})
        finally:
            torch.cuda.empty_cache()
            try:
                os.remove(path)
            except OSError:
                pass
    good = [e for e in out["entries"] if "h2d_GBps" in e]
    if good:
        best = max(good, key=lambda e: e["h2d_GBps"])
        out["roofline"] = {"bound": "pcie", "achieved": best["h2d_GBps"], "peak": PCIE_PEAK / 1e9, "unit": "GB/s", "frac": best["frac_of_pcie_peak"],
                           "note": "best whole-search H2D rate over the entries (rows %d, nq %d); the timed region starts with the rows in HOST memory "
                                   "(page cache), so this block is PCIe-inclusive by construction and is never the bench value" % (best["rows"], best["nq"])}
    return out


def cpu_baseline(db, q_unit, k, n_total, sd, embed_coords):
    """CPU legs on this host (baseline only).  `value`: the oracle's faiss-path port (oracle.c:orc_ip_topk, OpenMP +
    AVX2 FMA, all cores) on the first rows of the same database and the same query batch, scaled linearly in rows.
    `torch_cpu`: the reference's own torch op shapes (oracle/torch_baseline.py)."""
    from oracle import oracle as orc
    from oracle import torch_baseline as tb
    cores = orc.num_threads()
    qh = q_unit.cpu().numpy()
    sample = min(db.shape[0], 50_000)
    dbh = db[:sample].cpu().numpy()
    t = time.perf_counter(); orc.ip_topk(dbh, qh, k); dt = time.perf_counter() - t
    sample2 = int(min(db.shape[0], max(sample, sample * 5.0 / max(dt, 1e-3))))     # aim for ~5 s of wall time
    reps = 1
    if sample2 > sample:
        dbh = db[:sample2].cpu().numpy()
        t = time.perf_counter(); orc.ip_topk(dbh, qh, k); dt = time.perf_counter() - t
        sample = sample2
        reps = int(min(20, max(1, 2.0 / max(dt, 1e-3))))                           # whole database in well under a second: repeat
        if reps > 1:
            t = time.perf_counter()
            for _ in range(reps):
                orc.ip_topk(dbh, qh, k)
            dt = (time.perf_counter() - t) / reps
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    out = {"value": qh.shape[0] / (dt * n_total / sample), "unit": "queries/s", "cores": cores, "kind": "port",
           "sample": "first %d of %d rows x all %d queries, %.2f s wall per pass (%d passes), scaled linearly in rows" % (sample, n_total, qh.shape[0], dt, reps),
           "host": {"nproc": os.cpu_count(), "cpu_model": model}}
    rows_t = min(db.shape[0], 1_000_000)
    torch_legs = tb.time_search_legs(db[:rows_t].cpu().numpy(), qh, k, n_total)
    if sd is not None:
        torch_legs["egnn_batch1_loop"] = tb.time_egnn_leg(sd, embed_coords[:200], budget_s=10.0)
        # the C restatement of the encoder (oracle.c:orc_egnn_embed, OpenMP over edge rows) on the first structures of the same set
        from merizo_search_amd.foldclass import weights as W
        w, pe = W.pack_state_dict(sd)
        done, t0e = [], time.perf_counter()
        for c in embed_coords[:200]:
            orc.egnn_embed(w, pe, [c])
            done.append(len(c) ** 2)
            if time.perf_counter() - t0e > 5.0:
                break
        dte = time.perf_counter() - t0e
        total_sq = float(sum(len(c) ** 2 for c in embed_coords))
        out["oracle_egnn"] = {"embeds_per_s": len(embed_coords) / (dte * total_sq / float(sum(done))), "cores": cores, "kind": "port",
                              "sample": "%d of %d structures, %.2f s, scaled by sum N^2" % (len(done), len(embed_coords), dte)}
    out["torch_cpu"] = torch_legs
    return out


LINE_LIMIT = 4096          # bytes: the driver keeps a ~10 KB tail of the output; round 4's 22 KB line did not parse


def _r(x, nd=4):
    """Numbers of the compact line: 6 significant digits are more than any of them carries."""
    if isinstance(x, str):
        return x[:120]
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    return float("%.6g" % x)


def _roof_short(roof):
    """The contract's roofline object: bound / achieved / peak / unit / frac / traffic, plus the kernel it is about, its launch
    duration (HIP events, this run), the same work over the whole step, and where `traffic` came from."""
    if roof is None:
        return None
    out = {key: _r(roof.get(key)) for key in ("bound", "achieved", "peak", "unit", "frac", "step_frac", "traffic")}
    out["kernel"] = str(roof.get("kernel", ""))[:64]
    out["kernel_ms"] = _r(roof.get("kernel_ms"))
    for key in ("mfma_frac", "hbm_frac"):
        if roof.get(key) is not None:
            out[key] = _r(roof[key])
    if roof.get("traffic") is not None:
        out["traffic_from"] = str(roof.get("traffic_from", "profiles/ (committed rocprofv3 --pmc passes)"))[:96]
    return out


def compact_line(doc, full_path=None):
    """The ONE line the driver parses: the contract's keys, the top-level roofline, a short `prefiltered` summary, `cpu_baseline`
    and a handful of headline numbers of the other entries -- always under LINE_LIMIT bytes (tests/test_bench_line.py).  The full
    document (every block, every note) goes to `bench_full.json` beside this file."""
    cfg = doc.get("config", {})
    line = {key: _r(doc.get(key)) for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                              "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = str(line["metric"])[:120]
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200]}
    for key in ("db_rows", "rows_per_gpu", "dim", "queries_per_step", "k"):
        if key in cfg:
            line["config"][key] = cfg[key]
    line["config"]["path"] = "fp32 scan (ms_ip_topk), v_mfma_f32_32x32x2_f32"
    line["config"]["parallelism"] = str(cfg.get("sharding", ""))[:90]
    for key in ("recall_at_k", "planted_recall", "topk_identical_to_torch_bruteforce", "weak_scaling_ref_q_per_s", "vs_ref"):
        if doc.get(key) is not None:
            line[key] = _r(doc[key])
    co = doc.get("collective")
    if co:          # N > 1: what moved the bytes, between how many DISTINCT devices, and what the exchange + merge cost by itself
        line["collective"] = {"backend": _r(co.get("backend")), "world": co.get("world"), "distinct_devices": co.get("distinct_devices"),
                              "rccl_version": _r(co.get("rccl_version")), "exchange_us": _r(co.get("exchange_us")),
                              "same_device_selftest": co.get("same_device_selftest")}
    line["roofline"] = _roof_short(doc.get("roofline"))
    pf = doc.get("prefiltered")
    if pf:
        line["prefiltered"] = {"ms_per_step": _r(pf["ms_per_step"]), "queries_per_s": _r(pf["queries_per_s"]), "dtype": str(pf.get("dtype", ""))[:48],
                               "identical_to_fp32": pf.get("identical_to_fp32"), "exact_pass_queries": pf.get("exact_pass_queries"),
                               "roofline": _roof_short(pf.get("roofline"))}
    cb = doc.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": _r(cb["unit"]), "cores": cb["cores"], "kind": _r(cb["kind"]),
                                "sample": str(cb.get("sample", ""))[:140], "host": str((cb.get("host") or {}).get("cpu_model", ""))[:48]}
    # headline numbers of the other entries (everything else about them: the full document)
    more = {}
    c4 = doc.get("c4_shard")
    if c4:
        more["c4_shard"] = {"queries_per_s": _r(c4["queries_per_s"]), "ms_per_step": _r(c4["ms_per_step"]), "frac": _r(c4["roofline"]["frac"])}
        if c4.get("prefiltered"):
            more["c4_shard"]["prefiltered_queries_per_s"] = _r(c4["prefiltered"]["queries_per_s"])
            more["c4_shard"]["prefiltered_frac"] = _r(c4["prefiltered"]["roofline"]["frac"])
            more["c4_shard"]["prefiltered_identical"] = c4["prefiltered"].get("identical_to_fp32")
    if doc.get("c4_full"):
        more["c4_full"] = {key: _r(v) for key, v in doc["c4_full"].items() if key in ("rows", "queries_per_s", "ms_per_step", "frac", "identical_to_8_shards", "planted_recall")}
    hb = doc.get("hbm_regime")
    if hb:
        more["hbm_regime_step_frac"] = {"%dM_nq%d" % (round(e["rows"] / 1e6), e["nq"]): _r(float("%.3g" % e["step_frac_of_hbm_peak"])) for e in hb}
        # the same steps over the fp16 image (256 B per row): fraction of the HBM peak in IMAGE bytes, and the speed-up over the fp32 rows
        img = {"%dM_nq%d" % (round(e["rows"] / 1e6), e["nq"]): [_r(float("%.3g" % e["image"]["image_frac_of_hbm_peak"])), _r(float("%.3g" % (e["ms_per_step"] / e["image"]["ms_per_step"])))]
               for e in hb if e.get("image")}
        if img:
            more["hbm_regime_fp16_image_frac_and_speedup"] = img
    em = doc.get("embed")
    if em:
        more["embed"] = {"embeds_per_s": _r(em["embeds_per_s"]), "frac": _r(em["roofline"]["frac"])}
        if em.get("c5_query"):
            more["embed"]["c5_query_ms"] = _r(em["c5_query"].get("embed_ms_three_domains"))
    c3 = doc.get("c3_search")
    if c3:
        more["c3_search"] = {"ms_per_step": _r(c3["ms_per_step"]), "frac": _r(c3["roofline"]["frac"])}
        if c3.get("prefiltered"):
            more["c3_search"]["prefiltered_ms_per_step"] = _r(c3["prefiltered"]["ms_per_step"])
    if doc.get("c3_end_to_end"):
        more["c3_end_to_end_domains_per_s"] = _r(doc["c3_end_to_end"]["domains_per_s"])
    tw = doc.get("two_in_flight")
    if tw:
        more["two_in_flight"] = {"queries_per_s": _r(tw["queries_per_s"]), "ms_per_step": _r(tw["ms_per_step"]), "step_frac": _r(float("%.4g" % tw["step_frac"])),
                                 "identical": tw.get("identical_to_one_in_flight")}
        if tw.get("prefiltered"):
            more["two_in_flight"]["prefiltered_queries_per_s"] = _r(tw["prefiltered"]["queries_per_s"])
    stv = doc.get("streamed")
    if stv and stv.get("roofline"):
        ents = [e for e in stv["entries"] if "h2d_GBps" in e]
        big = max(ents, key=lambda e: (e["rows"], e["nq"] == 256))
        more["streamed"] = {"bound": "pcie", "h2d_GBps": _r(float("%.3g" % stv["roofline"]["achieved"])), "frac_of_63GBps": _r(float("%.3g" % stv["roofline"]["frac"])),
                            "rows": big["rows"], "nq": big["nq"], "rows_per_s": _r(float("%.3g" % big["rows_per_s"])),
                            "overlap_hidden_frac": _r(float("%.3g" % big["overlap_hidden_frac"])) if "overlap_hidden_frac" in big else None}
    cl = doc.get("clustered")
    if cl and cl.get("prefiltered"):
        more["clustered_prefiltered"] = {"queries_per_s": _r(cl["prefiltered"]["queries_per_s"]), "exact_pass_queries": cl["prefiltered"].get("exact_pass_queries"),
                                         "identical_to_fp32": cl["prefiltered"].get("identical_to_fp32")}
    if more:
        line["more"] = more
    if full_path:
        line["full"] = os.path.basename(full_path)
    text = json.dumps(line, separators=(",", ":"))
    # belt and braces: never over the limit, whatever the blocks above grew into
    for key in ("more", "prefiltered"):
        if len(text) < LINE_LIMIT:
            break
        line.pop(key, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:
        raise RuntimeError("bench line is %d bytes (limit %d)" % (len(text), LINE_LIMIT))
    return text


def write_full(doc):
    """The full document -> bench_full.json beside bench.py (the temp dir when that is read-only); -> path or None."""
    import tempfile
    for d in (REPO, tempfile.gettempdir()):
        try:
            path = os.path.join(d, "bench_full.json")
            with open(path, "w") as fh:
                json.dump(doc, fh, indent=1)
            return path
        except OSError:
            continue
    return None


PREFLIGHT_DEADLINE_S = 60.0


def _arm_deadline(seconds, what):
    """A rendezvous or a collective that hangs must cost a minute and name its line, not the driver's whole time limit: every rank dumps
    its Python stacks to stderr and EXITS when `seconds` pass before `_disarm_deadline` (faulthandler's watchdog thread: works while the
    main thread sits inside a C call)."""
    import faulthandler
    print("[bench] pre-flight: %s (deadline %.0f s)" % (what, seconds), file=sys.stderr, flush=True)
    faulthandler.dump_traceback_later(seconds, repeat=False, exit=True)


def _disarm_deadline():
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    if os.environ.get("MS_BENCH_FAULT_DUMP"):          # (the long-range diagnostic dump of the self-tests shares faulthandler's one timer)
        faulthandler.dump_traceback_later(float(os.environ["MS_BENCH_FAULT_DUMP"]), repeat=False, exit=False)


def collective_preflight(torch, dist, ops, sharded, dev, world, backend, same_device):
    """Before the ranks generate 23.4 GB each: prove that the process group works and say what it is.  One 8-byte all-reduce and one
    PackedExchange round trip (all-gather + ms_topk_merge_strided on a 4 x 3 result) under a 60 s deadline with the stack dump armed;
    an all-gather of every rank's PCI bus id -> the number of DISTINCT devices the ranks sit on.  backend "nccl" (= RCCL) with fewer
    distinct devices than ranks is refused (RCCL cannot share a device; a mis-set HIP_VISIBLE_DEVICES would otherwise only show as a hang
    or as N ranks timing one GPU).  -> the `collective` block of the N > 1 line (exchange_us is filled in after the timed region)."""
    _arm_deadline(PREFLIGHT_DEADLINE_S, "8-byte all-reduce + one PackedExchange round trip + PCI bus ids over %s, %d ranks" % (backend, world))
    rank = dist.get_rank()
    t = torch.full((1,), float(rank + 1), dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    assert float(t[0]) == world * (world + 1) / 2.0, "all-reduce returned %r" % float(t[0])
    ex = sharded.PackedExchange(4, 3, dev)
    ex.out_s.copy_(torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3).flip(1) + 100.0 * rank)      # rank r's lists: sorted, rows r*3..
    ex.out_i.copy_(torch.arange(3, dtype=torch.int64, device=dev)[None, :].repeat(4, 1) + 3 * rank)
    ex.exchange()
    ms_, mi_ = ex.merge()
    torch.cuda.synchronize()
    assert int(mi_[0, 0]) == 3 * (world - 1) and float(ms_[0, 0]) == 2.0 + 100.0 * (world - 1), "exchange + merge pre-flight returned a wrong list"
    bus = ops.device_pci_bus_id(dev)
    # (a fixed-size byte tensor on the device, gathered like the result blocks: no pickling, the same collective on either backend)
    tag = ("%s|%s" % (os.uname().nodename, bus)).encode()[:96].ljust(96, b"\0")
    mine = torch.tensor(list(tag), dtype=torch.uint8, device=dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    torch.cuda.synchronize()
    ids = [bytes(p_.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for p_ in parts]
    _disarm_deadline()
    distinct = len(set(ids))
    rccl = None
    try:
        v = torch.cuda.nccl.version()
        rccl = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass
    info = {"backend": dist.get_backend(), "world": world, "distinct_devices": distinct, "pci_bus_ids": sorted(set(i.split("|", 1)[1] for i in ids)),
            "rccl_version": rccl, "same_device_selftest": bool(same_device), "exchange_us": None,
            "preflight": "8-byte all-reduce + PackedExchange round trip passed within %.0f s" % PREFLIGHT_DEADLINE_S}
    refuse_shared_devices(dist.get_backend(), world, ids)
    return info


def refuse_shared_devices(backend, world, ids):
    """An RCCL run must have one DISTINCT device per rank (`ids`: one "host|pci bus id" string per rank): anything else is a mis-set
    HIP_VISIBLE_DEVICES / LOCAL_RANK that would show up as a hang, or as N ranks timing one GPU and calling it scaling.  gloo runs (the
    one-GPU self-tests, MS_BENCH_SAME_DEVICE=1) may share a device: their line says so (`distinct_devices`, `same_device_selftest`)."""
    distinct = len(set(ids))
    if backend == "nccl" and distinct != world:
        raise SystemExit("bench.py --gpus %d over RCCL: the %d ranks sit on %d distinct device(s) %s -- one GPU per rank is required "
                         "(check HIP_VISIBLE_DEVICES / LOCAL_RANK)" % (world, world, distinct, sorted(set(ids))))
    return distinct


def time_exchange(bench, reps=20):
    """The exchange step ALONE -- one all-gather of the packed per-shard results + the merge of the S blocks -- in microseconds, mean over
    `reps` back-to-back rounds between two fences, MAX over ranks: what the N-rank step adds to the one-rank step."""
    torch = bench.torch
    bench.ex.exchange(); bench.ex.merge(); bench.fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        bench.ex.exchange()
        bench.ex.merge()
    bench.fence()
    us = (time.perf_counter() - t0) / reps * 1e6
    t = torch.tensor([us], dtype=torch.float64, device=bench.db.device)
    bench.dist.all_reduce(t, op=bench.dist.ReduceOp.MAX)
    return float(t[0])


def pipelined_rate(bench, steps, depth=2):
    """Throughput with `depth` query batches in flight on `depth` HIP streams (each batch its own workspace, exchange block and outputs;
    the database, its image and the raw queries shared): the small launches of batch i + 1 (query preparation, sample pass, bound) and the
    merge of batch i - 1 run on the CUs the scan of batch i has not reached yet or has already left -- a serving loop's configuration.  NOT
    the bench value: with launches of two batches interleaved, the HIP-event duration of a scan launch includes the time it queues behind
    the other batch, so the top-level line (and its roofline) stays one batch in flight.  -> (ms per step, last results of every slot)."""
    torch = bench.torch
    dev = bench.db.device
    slots = [bench] + [bench.variant(bench.prefilter) for _ in range(depth - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    torch.cuda.synchronize()
    for st_ in streams:
        st_.wait_stream(torch.cuda.current_stream(dev))
    res = [None] * depth
    for i in range(4 * depth):
        with torch.cuda.stream(streams[i % depth]):
            res[i % depth] = slots[i % depth].step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % depth]):
            res[i % depth] = slots[i % depth].step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--rows", type=int, default=None, help="TOTAL database rows, sharded over the ranks (default: 1M at N=1; 45,625,000 PER GPU at N>1)")
    ap.add_argument("--nq", type=int, default=None, help="queries per step (default 256 at N=1, 4096 at N>1)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--shape", choices=("c2", "c4"), default=None,
                    help="N=1 only: c2 (default) = 1M rows x 256 queries; c4 = ONE rank's share of C4 (45,625,000 rows x 4096 queries) as the top-level "
                         "step -- the like-for-like first point of the weak-scaling curve that --gpus N>1 measures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip hbm_regime / c4_shard / embed")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child passes (N = 1, default C2 shape, extras on: two passes of ~25 s); the "
                         "figure of the committed profile is used instead")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the `two_in_flight` block (N = 1: the same steps with two query batches in flight on two HIP streams)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the `streamed` block (host memmap searched block by block over PCIe: writes up to 23.4 GB to /dev/shm)")
    ap.add_argument("--no-prefilter", action="store_true", help="skip the `prefiltered` blocks (the top-level line is the fp32 scan either way)")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="run the multi-GPU exchange + shard merge even on one GPU (validates that code path; slower)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without an outer launcher: start the N ranks as a FRESH child process tree
        # (torch.distributed.run, one rank per GPU) before anything in this process has touched the GPU, pass its
        # output through and exit with its code.  Nothing is exec'd; this parent never initialises HIP.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))

    if os.environ.get("MS_BENCH_FAULT_DUMP"):           # diagnostics: every rank dumps its Python stacks to stderr after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MS_BENCH_FAULT_DUMP"]), repeat=False, exit=False)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")

    from merizo_search_amd import _lib, ops
    from merizo_search_amd.foldclass import sharded, synthetic as syn

    _lib.require_gpu()                                   # fails loudly without the HIP library / a GPU
    # self-test hooks (one-GPU boxes): MS_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # MS_BENCH_BACKEND=gloo swaps the collective backend, so that the multi-rank logic can be run end to end
    same_device = os.environ.get("MS_BENCH_SAME_DEVICE") == "1"
    backend = os.environ.get("MS_BENCH_BACKEND", "nccl")                 # "nccl" is RCCL on ROCm
    if world > 1 and same_device and backend == "nccl":
        raise SystemExit("MS_BENCH_SAME_DEVICE=1 puts every rank on cuda:0, which RCCL cannot do: set MS_BENCH_BACKEND=gloo")
    dev_index = 0 if same_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    collective = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _arm_deadline(2 * PREFLIGHT_DEADLINE_S, "rendezvous of %d ranks (%s)" % (world, backend))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        _disarm_deadline()
        collective = collective_preflight(torch, dist, ops, sharded, dev, world, backend, same_device)

    k = args.k
    weak = world > 1 and args.rows is None
    if world > 1 and args.shape == "c2":
        raise SystemExit("--shape c2 is the one-GPU workload; --gpus N > 1 always runs the C4 shape (weak scaling)")
    if world == 1 and args.shape == "c4":
        n_total, nq = args.rows or int(os.environ.get("MS_BENCH_ROWS_PER_GPU", C4_ROWS_PER_GPU)), args.nq or C4_NQ
        if args.steps == ap.get_default("steps"):
            args.steps, args.warmup = 10, 2                               # 0.34 s per step
    elif world == 1:
        n_total, nq = args.rows or 1_000_000, args.nq or 256
    else:
        rows_per_gpu = int(os.environ.get("MS_BENCH_ROWS_PER_GPU", C4_ROWS_PER_GPU))
        n_total, nq = args.rows or rows_per_gpu * world, args.nq or C4_NQ
    lo, hi = sharded.shard_bounds(n_total, world, rank)
    log = (lambda m: print("[bench] " + m, file=sys.stderr, flush=True)) if rank == 0 else (lambda m: None)

    use_pf = not args.no_prefilter
    bench = SearchBench(torch, dist, ops, syn, sharded, dev, rank, world, n_total, lo, hi, nq, k, exchange=args.exercise_exchange)
    steps = args.steps
    elapsed, scan_ms, res = bench.run(steps, args.warmup)        # the fp32 scan: the reference's arithmetic, data-independent
    checks = bench.check(res, sharded)
    # N > 1 (weak scaling): the same ranks' shard scans without the exchange -- the curve's one-rank reference from the same build
    local_ref = bench.local_rate(max(2, min(steps, 10))) if world > 1 else None
    if collective is not None:
        collective["exchange_us"] = time_exchange(bench)

    def pf_block(b_fp32, res_fp32, steps_, warm_, prep_s=0.3, pmc=None):
        """The same step through the prefiltered search (the path the driver takes for this shape)."""
        bp = b_fp32.variant(True)
        if not bp.prefilter:
            return None
        torch.cuda.synchronize()
        el, sc, rp = bp.run(steps_, warm_, prep_budget_s=prep_s)
        ms = el / steps_ * 1e3
        fmt = pf_format_name(ops, bp.image)
        blk = {"ms_per_step": ms, "queries_per_s": bp.nq / ms * 1e3, "dtype": PF_FORMATS[fmt][3],
               "identical_to_fp32": bool(torch.equal(rp[1], res_fp32[1]) and torch.equal(rp[0].view(torch.int32), res_fp32[0].view(torch.int32))),
               "exact_pass_queries": ops.prefilter_flagged(bp.ws) if world == 1 else None,
               "image_bytes": int(bp.image.numel()),
               "roofline": roofline(bp.nq, bp.n_local, bp.k, sc, ms, prefiltered=fmt),
               "note": "see notes.prefiltered"}
        if pmc is not None:
            attach_committed_traffic(blk["roofline"], pmc)
        del bp
        return blk

    c2 = (n_total, nq, k, world) == (1_000_000, 256, 10, 1)
    pf_main = pf_block(bench, res, max(20, min(steps, 200)), 10, pmc="r06_pf_c2_pmc.json" if c2 else None) if use_pf else None

    two = None
    if world == 1 and nq > 64 and not args.no_pipelined:
        # the same steps with two batches in flight (informative: see pipelined_rate)
        ms2, r2 = pipelined_rate(bench, max(20, min(steps, 200)))
        same = all(bool(torch.equal(r_[1], res[1]) and torch.equal(r_[0].view(torch.int32), res[0].view(torch.int32))) for r_ in r2)
        two = {"depth": 2, "ms_per_step": ms2, "queries_per_s": nq / ms2 * 1e3, "identical_to_one_in_flight": same,
               "step_frac": roofline(nq, bench.n_local, k, scan_ms, ms2)["step_frac"],
               "note": "two query batches in flight on two HIP streams (own workspaces): the small launches of one batch fill the CUs the other's scan leaves idle in "
                       "its ramp and tail; not the bench value (launch durations measured by events would include queueing)"}
        if use_pf:
            bp2 = bench.variant(True)
            if bp2.prefilter:
                msp2, rp2 = pipelined_rate(bp2, max(20, min(steps, 200)))
                two["prefiltered"] = {"ms_per_step": msp2, "queries_per_s": nq / msp2 * 1e3,
                                      "identical_to_fp32": all(bool(torch.equal(r_[1], res[1]) and torch.equal(r_[0].view(torch.int32), res[0].view(torch.int32))) for r_ in rp2)}
            del bp2
        if rank == 0:
            print("[bench] two batches in flight: fp32 %.4f ms per step = %.0f q/s%s" % (ms2, nq / ms2 * 1e3,
                  "; prefiltered %.4f ms = %.0f q/s" % (two["prefiltered"]["ms_per_step"], two["prefiltered"]["queries_per_s"]) if "prefiltered" in two else ""), file=sys.stderr, flush=True)
    if rank == 0:
        ms_per_step = elapsed / steps * 1e3
        roof = roofline(nq, bench.n_local, k, scan_ms, ms_per_step)
        if c2:
            attach_committed_traffic(roof, "r06_c2_pmc.json")
        if (n_total, nq, world) == (1_000_000, 256, 1):
            workload = "C2: brute-force cosine top-%d, 1M x 128 fp32 synthetic DB, batch=256 queries, 1 MI355X" % k
        elif world == 1 and args.shape == "c4":
            workload = "C4 shape, ONE rank's share: %d x 128 fp32 synthetic DB rows, batch=%d queries, top-%d, 1 MI355X (the N=1 point of the weak-scaling curve)" % (n_total, nq, k)
        elif weak:
            workload = "C4 shape: TED-scale %d x 128 fp32 synthetic DB row-sharded over %d GPUs (%d rows each), batch=%d queries, top-%d, RCCL all-gather of per-shard top-k" % (
                n_total, world, bench.n_local, nq, k)
        else:
            workload = "cosine top-%d, %d x 128 fp32 synthetic DB, batch=%d queries, %d GPU(s)" % (k, n_total, nq, world)
        line = {
            "metric": "queries/sec, exact 128-d cosine top-%d (recall@%d vs brute force = %.4f)" % (k, k, checks["recall_at_k"]),
            "value": nq * steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "db_rows": n_total, "rows_per_gpu": bench.n_local, "dim": 128, "queries_per_step": nq, "k": k,
                       "score": "inner product of L2-normalised rows (normalisation of the query batch inside the step)",
                       "path": "fp32 scan (ms_ip_topk_prepare / _scan / _finish): v_mfma_f32_32x32x2_f32, the reference's arithmetic; the prefiltered "
                               "search of the same step is the block `prefiltered`",
                       "sharding": "contiguous row shards, one RCCL all-gather of per-shard top-k + merge per step" if world > 1 else "single shard",
                       "scaling_note": "weak: rows per GPU fixed, the database grows with N, so ideal queries/s is CONSTANT in N (row x query "
                                       "rate grows N-fold); compare with c4_shard of the N=1 run" if weak else None},
            "row_queries_per_s": float(n_total) * nq * steps / elapsed,
            "weak_scaling_ref_q_per_s": local_ref,
            "vs_ref": (nq * steps / elapsed) / local_ref if local_ref else None,
            "weak_scaling_ref_note": ("the same ranks' shard scans timed WITHOUT the all-gather and the shard merge (slowest rank): the one-GPU rate on one "
                                      "45.6M-row share; weak scaling holds queries/s constant while the database grows N-fold, so vs_ref is the scaling "
                                      "efficiency against this build's own one-rank point (`python bench.py --gpus 1 --shape c4` measures it alone)") if local_ref else None,
            "collective": collective,
            "two_in_flight": two,
            "roofline": roof,
            "prefiltered": pf_main,
        }
        line.update(checks)
        if pf_main is not None:
            log("prefiltered: %.4f ms per step = %.0f q/s, scan %.1f us, identical to the fp32 scan: %s, exact-pass queries: %s" % (
                pf_main["ms_per_step"], pf_main["queries_per_s"], pf_main["roofline"]["kernel_ms"] * 1e3, pf_main["identical_to_fp32"], pf_main["exact_pass_queries"]))
    else:
        line = None

    extras_sd, extras_coords = None, None
    if world == 1 and not args.no_extras and args.shape != "c4":
        db_keep, q_keep = bench.db, bench.q_raw
        del bench.ws
        mk = lambda rows, nq_, **kw_: SearchBench(torch, dist, ops, syn, sharded, dev, 0, 1, rows, 0, rows, nq_, k, **kw_)
        if use_pf and c2:
            # C2 on clustered data: 64 of the 256 queries own a family of 200 near-duplicate rows
            bc = mk(n_total, nq, clustered=64)
            elc, scc, rc_ = bc.run(60, 10, prep_budget_s=0.1)
            msc = elc / 60 * 1e3
            pfc = pf_block(bc, rc_, 60, 10, prep_s=0.1)
            line["clustered"] = {"workload": "C2 with 64 of the 256 queries owning a family of 200 rows within ~1e-6 of each other (no proof possible for them)",
                                 "fp32_path": {"ms_per_step": msc, "queries_per_s": nq / msc * 1e3, "scan_ms": scc},
                                 "prefiltered": pfc}
            log("clustered: fp32 %.4f ms per step; prefiltered %.4f ms per step, %s queries through the exact pass, identical: %s" % (
                msc, pfc["ms_per_step"], pfc["exact_pass_queries"], pfc["identical_to_fp32"]))
            del bc, rc_
            torch.cuda.empty_cache()
        line["hbm_regime"] = hbm_regime(mk, (1_000_000, 4_000_000), log)
        free, _tot = torch.cuda.mem_get_info(dev)
        if free > 70 << 30:
            big = mk(C4_ROWS_PER_GPU, C4_NQ)
            el, sc, r4 = big.run(2, 1, prep_budget_s=0.0)
            ms4 = el / 2 * 1e3
            fi4 = r4[1].cpu()
            planted4 = float(np.mean([len(set(big.planted[j].tolist()) & set(fi4[j].tolist())) / 3.0 for j in range(C4_NQ)]))
            line["c4_shard"] = {"workload": "one rank's share of C4: %d x 128 rows x %d queries, top-%d" % (C4_ROWS_PER_GPU, C4_NQ, k),
                                "ms_per_step": ms4, "queries_per_s": C4_NQ / ms4 * 1e3, "planted_recall": planted4,
                                "note": "queries_per_s here = the N-GPU rate on an N x 45.6M-row database, minus the all-gather + merge of 480 KB per rank "
                                        "(a projection: no multi-GPU node was available to this build)",
                                "roofline": roofline(C4_NQ, C4_ROWS_PER_GPU, k, sc, ms4)}
            if k == 10:
                attach_committed_traffic(line["c4_shard"]["roofline"], "r06_c4_pmc.json")
            log("c4_shard: %.1f ms per 4096-query batch = %.0f q/s (fp32 scan %.1f ms = %.1f%% of fp32 MFMA peak)" % (ms4, C4_NQ / ms4 * 1e3, sc, line["c4_shard"]["roofline"]["frac"] * 100))
            if use_pf:
                line["c4_shard"]["prefiltered"] = pf_block(big, r4, 2, 1, prep_s=0.0, pmc="r06_pf_c4_pmc.json" if k == 10 else None)
                p4 = line["c4_shard"]["prefiltered"]
                log("c4_shard prefiltered: %.1f ms per batch = %.0f q/s, scan %.1f ms, identical: %s" % (p4["ms_per_step"], p4["queries_per_s"], p4["roofline"]["kernel_ms"], p4["identical_to_fp32"]))
            # the HBM-bound regime on the same 23.4 GB shard: reuse its rows
            small = []
            for nq_ in (1, 4, 8, 32):
                b = SearchBench.__new__(SearchBench)
                b.__dict__.update(big.__dict__)
                b.nq, b.q_raw = nq_, big.q_raw[:nq_].contiguous()
                b.q = torch.empty_like(b.q_raw)
                b.prefilter = False
                b.ws = torch.empty_like(ops.TopKWorkspace(dev).get(b.n_local, nq_, k))
                b.ex = sharded.PackedExchange(nq_, k, dev)
                el, sc, _ = b.run(6, 2, prep_budget_s=0.0)
                ms = el / 6 * 1e3
                small.append({"rows": C4_ROWS_PER_GPU, "nq": nq_, "k": k, "ms_per_step": ms, "scan_ms": sc, "queries_per_s": nq_ / ms * 1e3,
                              "scan_GBps": 512.0 * C4_ROWS_PER_GPU / sc / 1e6, "scan_frac_of_hbm_peak": 512.0 * C4_ROWS_PER_GPU / (sc * 1e-3) / HBM_PEAK,
                              "step_frac_of_hbm_peak": 512.0 * C4_ROWS_PER_GPU / (ms * 1e-3) / HBM_PEAK, "kernel": scan_kernel_name(nq_, k),
                              "note": "see notes.small_batch.nq%d" % nq_})
                log("hbm_regime rows=%d nq=%d: scan %.3f ms (%.1f%% of 8 TB/s)" % (C4_ROWS_PER_GPU, nq_, sc, small[-1]["scan_frac_of_hbm_peak"] * 100))
                if use_pf:
                    few_image_entry(b, small[-1], 6, log)
                del b
            line["hbm_regime"] += small
            del big, r4
            torch.cuda.empty_cache()
        free, _tot = torch.cuda.mem_get_info(dev)
        if free > 200 << 30:
            # C4 at its real row count on ONE GPU: all 365,000,000 rows (186.9 GB) resident, one unsharded scan per 4096-query batch
            # -- and the same rows as eight sequential shard scans + the strided merge, which is what eight ranks compute
            n_full, S_ = 8 * C4_ROWS_PER_GPU, 8
            full = mk(n_full, C4_NQ)
            el, sc, rf = full.run(2, 1, prep_budget_s=0.0)
            msf = el / 2 * 1e3
            idx_off = (4 * C4_NQ * k + 7) // 8 * 8
            gathered = torch.zeros((S_, idx_off + 8 * C4_NQ * k), dtype=torch.uint8, device=dev)
            qn_ = full.q_raw / full.q_raw.norm(dim=1, keepdim=True)
            ops.l2_normalize_rows(full.q_raw, 1e-12, out=full.q)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for r_ in range(S_):
                lo_, hi_ = sharded.shard_bounds(n_full, S_, r_)
                out_ = (gathered[r_, : 4 * C4_NQ * k].view(torch.float32).reshape(C4_NQ, k), gathered[r_, idx_off:].view(torch.int64).reshape(C4_NQ, k))
                ops.ip_topk(full.db[lo_:hi_], full.q, k, row_offset=lo_, out=out_)
            ms8, mi8 = torch.empty_like(rf[0]), torch.empty_like(rf[1])
            ops.topk_merge_packed(gathered, S_, C4_NQ, k, idx_off, ms8, mi8)
            torch.cuda.synchronize(); t8 = (time.perf_counter() - t0) * 1e3
            fif = rf[1].cpu()
            line["c4_full"] = {"workload": "C4 at its real size on ONE GPU: %d x 128 fp32 rows (%.1f GB) resident, %d queries, top-%d, one unsharded fp32 scan per batch" % (
                                   n_full, n_full * 512 / 1e9, C4_NQ, k),
                               "rows": n_full, "ms_per_step": msf, "queries_per_s": C4_NQ / msf * 1e3,
                               "frac": roofline(C4_NQ, n_full, k, sc, msf)["frac"], "roofline": roofline(C4_NQ, n_full, k, sc, msf),
                               "eight_sequential_shards_plus_merge_ms": t8,
                               "identical_to_8_shards": bool(torch.equal(mi8, rf[1]) and torch.equal(ms8.view(torch.int32), rf[0].view(torch.int32))),
                               "planted_recall": float(np.mean([len(set(full.planted[j].tolist()) & set(fif[j].tolist())) / 3.0 for j in range(C4_NQ)]))}
            log("c4_full: %.1f ms per 4096-query batch over 365M rows on one GPU = %.0f q/s (scan %.1f%% of fp32 MFMA peak); 8 shards + merge %.1f ms, identical: %s" % (
                msf, C4_NQ / msf * 1e3, line["c4_full"]["frac"] * 100, t8, line["c4_full"]["identical_to_8_shards"]))
            del full, rf, gathered, ms8, mi8, qn_
            torch.cuda.empty_cache()
        # list length: the same C2 shape at other k (lists of 5 / 10 / 16 / 32 entries per lane for k <= 10 / 20 / 32 / 64)
        line["k_sweep"] = []
        for kk in (1, 10, 20, 32, 64):
            bk = SearchBench(torch, dist, ops, syn, sharded, dev, 0, 1, n_total, 0, n_total, nq, kk)
            el, sc, rk = bk.run(40, 5, prep_budget_s=0.05)
            ent = {"k": kk, "ms_per_step": el / 40 * 1e3, "scan_ms": sc, "queries_per_s": nq * 40 / el, "kernel": scan_kernel_name(nq, kk)}
            if use_pf:
                bp = bk.variant(True)
                if bp.prefilter:
                    elp, scp, rp = bp.run(40, 5, prep_budget_s=0.05)
                    ent["prefiltered"] = {"ms_per_step": elp / 40 * 1e3, "scan_ms": scp,
                                          "identical_to_fp32": bool(torch.equal(rp[1], rk[1]) and torch.equal(rp[0].view(torch.int32), rk[0].view(torch.int32)))}
                del bp
            line["k_sweep"].append(ent)
            log("k_sweep k=%d: fp32 %.3f ms per step (scan %.3f ms)%s" % (kk, el / 40 * 1e3, sc, "; prefiltered %.3f ms" % ent["prefiltered"]["ms_per_step"] if "prefiltered" in ent else ""))
            del bk
        line["c3_search"], c3_state = c3_search_bench(torch, ops, syn, dev, k, log, prefilter=use_pf)
        line["embed"], extras_sd, extras_coords, enc, lens = embed_bench(torch, ops, log)
        line["c3_end_to_end"] = c3_end_to_end(torch, ops, enc, extras_coords, lens, c3_state, log)
        del c3_state, enc
        torch.cuda.empty_cache()
        if not args.no_streamed:
            line["streamed"] = streamed_bench(torch, ops, syn, dev, k, log)
        bench.db, bench.q_raw = db_keep, q_keep

    if rank == 0 and world == 1 and c2 and not args.no_live_traffic and (not args.no_extras or os.environ.get("MS_BENCH_LIVE_TRAFFIC") == "1"):
        # roofline.traffic of the top-level kernel, measured in THIS run (the GPU is idle now: every timed region is over)
        # (the prefiltered step's image scan is in the same child runs: both kernels' counters from the same two passes)
        live = measure_traffic_live({"top": "ms_scan_loader_kernel<5, 0, false, false>", "pf": "ms_scan_pf16_kernel<10, 8, false, false,"}, log)
        targets = {"top": line["roofline"], "pf": (line.get("prefiltered") or {}).get("roofline")}
        for key, roof_ in targets.items():
            if live is None or key not in live or roof_ is None:
                continue
            roof_["traffic_committed_profile"] = roof_.get("traffic")
            roof_["traffic"] = live[key][0]
            roof_["traffic_from_committed_profile"] = False
            roof_["traffic_from"] = "measured in this run (2 rocprofv3 --pmc child passes, %d launches)" % live[key][1]
            roof_["traffic_source"] = roof_["traffic_from"]
            log("live traffic of %s: %.1f MB = %.3f x the algorithmic bytes" % (roof_["kernel"], live[key][0] / 1e6, live[key][0] / roof_["algorithmic_bytes_per_launch"]))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            q_unit = bench.q_raw / bench.q_raw.norm(dim=1, keepdim=True)
            line["cpu_baseline"] = cpu_baseline(bench.db, q_unit, k, n_total, extras_sd, extras_coords)
        line["notes"] = {
            "prefiltered": "ms_ip_topk_prefiltered over the image built when the database became resident (fp16 rows, 256 B per row next to the "
                           "fp32 rows; MS_PF_F16X2: 2 fp16 matrix instructions per 16 dimensions); the 2k best rows per query re-scored with the exact fp32 chain, per-query proof of completeness, an exact "
                           "fp32 pass for the queries whose proof failed (exact_pass_queries of them); results bit-identical to the fp32 scan "
                           "(tests/test_prefilter_gpu.py)",
            "prefiltered_roofline": "achieved / frac (matrix-bound shapes) count the flops the launch EXECUTES: m 16-bit matrix instructions per 16 "
                                    "dimensions = m x the algorithmic flops (f16x2: 2, f16x1: 1, bf16x3: 3), against the dense 16-bit matrix peak -- the roof of this arithmetic; "
                                    "algorithmic_tflops is the un-tripled figure (it may exceed the fp32 matrix peak: this is not fp32 matrix work)",
            "traffic": "roofline.traffic = HBM bytes per launch from the rocprofv3 --pmc passes named in traffic_source (gfx950 corrections: "
                       "tools/pmc_to_json.py); counters need the profiler, so they are not collected inside this run",
            "small_batch": {"nq%d" % q_: small_batch_note(q_, ops) for q_ in (1, 4, 8, 32)},
        }
        full_path = write_full(line)
        print(json.dumps(line), file=sys.stderr, flush=True)                 # the full document: stderr and bench_full.json
        log("full document: %s" % full_path)
        sys.stderr.flush()
        print(compact_line(line, full_path), flush=True)                     # the ONE stdout line (< 4 KB)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
