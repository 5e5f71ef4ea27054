"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch, per kernel (non-counting builds only)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "<true>" in k or "hrt_" not in k:
            continue
        name = "path" if "path_trace" in k else ("primary" if "primary" in k else k[:20])
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print("kernel", k)
    for c, v in sorted(d.items()):
        print("  %-34s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
