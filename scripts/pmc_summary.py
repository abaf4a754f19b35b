"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel (per dispatch)."""
import csv, sys, glob, collections, os
def short(n):
    n = n.split("(")[0].replace("void ", "").replace("rgck::", "")
    return n[:40]
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in acc.items():
            print(f"{k:42s}", "  ".join(f"{c}={sum(v)/len(v):.4g}(x{len(v)})" for c, v in sorted(cs.items())))
