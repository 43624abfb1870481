"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            if "ms_" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(d.split("/")[-1], k)
            for c, x in sorted(v.items()):
                print("    %-28s %16.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
