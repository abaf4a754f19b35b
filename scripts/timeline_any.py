"""Kernel timeline of the last `frames` frames in a rocprofv3 --kernel-trace csv, a frame starting at kernel name `marker`."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
start, end = idx[-2], idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
busy = 0
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rgck::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q={r.get('Queue_Id', '?'):>3}  {name}")
print("frame span us:", (int(rows[end]["Start_Timestamp"]) - t0) / 1e3, " kernel-busy us:", busy / 1e3)
