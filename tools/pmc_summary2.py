import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "true>" in k or "hrt_" not in k:
            continue
        name = k.split("(")[0].replace("void ", "").replace("hrt::", "")[:48]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    print("kernel", k)
    print("   " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
