"""The ONE stdout line of bench.py must stay parseable by the driver: one JSON object under 4 KB carrying the contract's keys, the
top-level `roofline` and `cpu_baseline` (round 4's line was 22 KB and the driver's tail cut its head off).  CPU tests: the line
builder on canned documents; the GPU suite asserts the same on real output (tests/test_sharded_drivers.py:_bench_line,
tests/test_drivers_gpu.py)."""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (importing bench.py touches neither torch nor the GPU)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _doc():
    with open(os.path.join(REPO, "tests", "golden", "bench_doc_r04.json")) as fh:
        return json.load(fh)


def _check(text):
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 4096
    line = json.loads(text)
    for key in CONTRACT:
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "step_frac"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert set(line["config"]) >= {"workload", "db_rows", "queries_per_step", "k"} and "model" not in line["config"]
    return line


def test_line_of_the_round_4_document_is_small_and_complete():
    """The very document whose 22 KB line the driver could not parse."""
    doc = _doc()
    assert len(json.dumps(doc)) > 20000
    line = _check(bench.compact_line(doc, "/somewhere/bench_full.json"))
    assert line["value"] == float("%.6g" % doc["value"]) and line["dtype"] == "f32" and line["config"]["workload"].startswith("C2:")
    assert abs(line["roofline"]["frac"] - doc["roofline"]["frac"]) < 1e-5 and line["roofline"]["bound"] == "mfma"
    assert line["prefiltered"]["identical_to_fp32"] is True and line["prefiltered"]["roofline"]["kernel_ms"] > 0
    assert line["more"]["c4_shard"]["queries_per_s"] > 1e4 and line["full"] == "bench_full.json"
    assert line["cpu_baseline"]["cores"] == 128 and line["cpu_baseline"]["kind"] == "port"


def test_line_stays_under_the_limit_whatever_the_blocks_say():
    """Every free-text field blown up to 5,000 characters, 40 hbm_regime entries: still under the limit, contract keys intact."""
    doc = _doc()
    long = "x" * 5000

    def blow(x):
        if isinstance(x, dict):
            return {k: blow(v) for k, v in x.items()}
        if isinstance(x, list):
            return [blow(v) for v in x]
        return long if isinstance(x, str) and x not in ("mfma", "hbm", "TFLOP/s", "GB/s", "queries/s", "port", "f32", "synthetic", "weak") else x

    big = blow(copy.deepcopy(doc))
    big["hbm_regime"] = big["hbm_regime"] * 4
    big["weak_scaling_ref_q_per_s"], big["vs_ref"] = 12345.678, 0.98765
    line = _check(bench.compact_line(big, None))
    assert line["vs_ref"] == 0.98765


def test_line_without_optional_blocks():
    """--no-extras --no-prefilter --no-cpu-baseline, N > 1: only the contract keys remain (cpu_baseline is rank 0 at N = 1 only)."""
    doc = {k: v for k, v in _doc().items() if k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                                     "vs_baseline", "dtype", "data", "config", "roofline", "recall_at_k", "planted_recall")}
    doc["n_gpus"] = 8
    text = bench.compact_line(doc)
    line = json.loads(text)
    assert len(text) < 2048 and line["n_gpus"] == 8 and "cpu_baseline" not in line and "prefiltered" not in line and "more" not in line
    assert "collective" not in line
    # an N > 1 document says what moved the bytes (round 6): the line carries it
    doc["collective"] = {"backend": "nccl", "world": 8, "distinct_devices": 8, "pci_bus_ids": ["0000:%02x:00.0" % (5 + 16 * i) for i in range(8)],
                         "rccl_version": "2.22.3", "same_device_selftest": False, "exchange_us": 61.25, "preflight": "passed"}
    line = json.loads(bench.compact_line(doc))
    assert line["collective"] == {"backend": "nccl", "world": 8, "distinct_devices": 8, "rccl_version": "2.22.3", "exchange_us": 61.25,
                                  "same_device_selftest": False}


def test_rccl_ranks_that_share_a_device_are_refused():
    """Round 6: an N > 1 run over RCCL whose ranks do not sit on N distinct devices fails before anything is timed; gloo
    self-tests on one device pass (and their line says `distinct_devices: 1`)."""
    import pytest
    eight = ["node|0000:%02x:00.0" % (5 + 16 * i) for i in range(8)]
    assert bench.refuse_shared_devices("nccl", 8, eight) == 8
    assert bench.refuse_shared_devices("gloo", 8, ["node|0000:05:00.0"] * 8) == 1
    with pytest.raises(SystemExit, match="distinct device"):
        bench.refuse_shared_devices("nccl", 8, eight[:7] + eight[:1])
    with pytest.raises(SystemExit, match="one GPU per rank"):
        bench.refuse_shared_devices("nccl", 2, ["node|0000:05:00.0"] * 2)


def test_line_carries_the_streamed_headline():
    """Round 6: the `streamed` block (host memmap over PCIe) puts its two headline numbers under `more`."""
    doc = _doc()
    doc["streamed"] = {"bound": "pcie", "peak_GBps": 63.0, "roofline": {"bound": "pcie", "achieved": 38.123, "peak": 63.0, "unit": "GB/s", "frac": 0.6051},
                       "entries": [{"rows": 8_000_000, "nq": 256, "h2d_GBps": 38.123, "rows_per_s": 7.4e7, "frac_of_pcie_peak": 0.6051, "overlap_hidden_frac": 0.93},
                                   {"rows": 45_625_000, "nq": 1, "h2d_GBps": 37.0, "rows_per_s": 7.2e7, "frac_of_pcie_peak": 0.587, "overlap_hidden_frac": 0.9},
                                   {"rows": 45_625_000, "nq": 256, "h2d_GBps": 36.5, "rows_per_s": 7.1e7, "frac_of_pcie_peak": 0.579, "overlap_hidden_frac": 0.95},
                                   {"rows": 45_625_000, "skipped": "x"}]}
    line = _check(bench.compact_line(doc, None))
    st = line["more"]["streamed"]
    assert st["bound"] == "pcie" and st["h2d_GBps"] == 38.1 and st["rows"] == 45_625_000 and st["nq"] == 256 and st["overlap_hidden_frac"] == 0.95


def test_write_full_round_trips(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    doc = _doc()
    path = bench.write_full(doc)
    assert path == str(tmp_path / "bench_full.json")
    with open(path) as fh:
        assert json.load(fh) == doc
