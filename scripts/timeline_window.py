"""Kernel timeline (start offset, duration, queue) of a window of a rocprofv3 --kernel-trace csv: the last `us` microseconds before the
last kernel, default 900."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 900.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0   # stop this long before the end (the drain is not steady state)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"]) - int(skip * 1e3)
t0 = t_end - int(win * 1e3)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t_end: continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rgck::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q={r.get('Queue_Id', '?'):>3}  {name}")
