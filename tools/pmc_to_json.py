"""rocprofv3 --pmc passes -> one JSON summary per kernel:
    python tools/pmc_to_json.py OUT.json "<kernel name substring>" "<workload text>" <algorithmic bytes per launch> <algorithmic flops per launch> DIR [DIR...]
Each DIR holds one `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` pass of the same command (counters that
cannot share a pass go in separate runs: FETCH_SIZE; WRITE_SIZE; GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES).
Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count KiB; on gfx950 FETCH_SIZE
reports half of a wide coalesced / LDS-DMA read stream, so HBM read bytes = 2 x FETCH_SIZE x 1024."""
import collections, csv, glob, json, sys

out, kern, workload, alg_bytes, alg_flops = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), float(sys.argv[5])
vals = collections.defaultdict(list)
for d in sys.argv[6:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"kernel": kern, "workload": workload, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": alg_flops}
for c, x in sorted(vals.items()):
    res[c] = {"mean_per_launch": sum(x) / len(x), "launches": len(x)}
if "FETCH_SIZE" in res:
    res["hbm_read_bytes_corrected"] = 2.0 * res["FETCH_SIZE"]["mean_per_launch"] * 1024
if "WRITE_SIZE" in res:
    res["hbm_write_bytes_reported"] = res["WRITE_SIZE"]["mean_per_launch"] * 1024
if "hbm_read_bytes_corrected" in res:
    res["traffic_bytes_per_launch"] = res["hbm_read_bytes_corrected"] + res.get("hbm_write_bytes_reported", 0.0)
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / alg_bytes if alg_bytes else None
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
    # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
    res["matrix_pipe_busy_fraction"] = (res["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0) / (res["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not isinstance(v, dict)}, indent=1))
