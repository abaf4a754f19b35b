"""Print the kernel timeline of the last registration frame in a rocprofv3 --kernel-trace csv (start offsets in us)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame = from the last k_bbox pair backwards: find last two k_bbox
idx = [i for i, r in enumerate(rows) if "k_bbox" in r["Kernel_Name"]]
start = idx[-2]
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rgck::", "")[:34]
    q = r.get("Queue_Id", "?")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q={q:>3}  {name}")
print("frame span us:", (int(rows[-1]["End_Timestamp"]) - t0) / 1e3)
