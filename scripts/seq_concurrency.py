"""What stops S sequences on one GPU from adding up: from a rocprofv3 --kernel-trace of rgc-slam_amd/cpp/sequences_per_gpu (S sequences
at once), over the LAST `window` fraction of the trace (the concurrent passes):
  * how much of the wall time had 0 / 1 / 2 / 3+ kernels in flight,
  * per kernel: launches, average duration in the window, and the same kernel's average duration in the FIRST part of the trace (every
    sequence is first run alone there) -- a kernel that takes S times longer beside S - 1 others was using the whole chip alone,
  * the chip-filling time per frame: the sum of the alone-durations of the kernels whose launches fill every CU (>= 1024 workgroups).
    python scripts/seq_concurrency.py <rocprof output dir> <S> [window, default 0.5]"""
import collections, csv, glob, json, sys
d, S = sys.argv[1], int(sys.argv[2])
window = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rgck::", "")
    wg = int(r.get("Workgroup_Size_X", 0) or 0)
    gx = int(r.get("Grid_Size_X", 0) or 0)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, gx // max(wg, 1)))
rows.sort()
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
cut = t_hi - int((t_hi - t_lo) * window)
alone_end = t_lo + int((t_hi - t_lo) * (1.0 - window) * 0.9)
conc = [r for r in rows if r[0] >= cut]
alone = [r for r in rows if r[1] <= alone_end]
# time with n kernels in flight
ev = []
for s, e, _, _ in conc:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist, cur, last = collections.Counter(), 0, cut
for t, dlt in ev:
    hist[min(cur, 3)] += t - last
    last, cur = t, cur + dlt
tot = sum(hist.values())
def stats(rs):
    acc = collections.defaultdict(lambda: [0, 0, 0])
    for s, e, n, wgs in rs:
        a = acc[n]; a[0] += 1; a[1] += e - s; a[2] = max(a[2], wgs)
    return acc
sa, sc = stats(alone), stats(conc)
out = {"S": S, "window_ms": round((t_hi - cut) / 1e6, 2), "time_with_n_kernels_in_flight": {str(k) + ("+" if k == 3 else ""): round(v / tot, 4) for k, v in sorted(hist.items())},
       "kernels": []}
for n, (c, t, wgs) in sorted(sc.items(), key=lambda kv: -kv[1][1])[:12]:
    a = sa.get(n)
    out["kernels"].append({"kernel": n[:48], "workgroups": wgs, "avg_us_beside_others": round(t / c / 1e3, 1), "avg_us_alone": round(a[1] / a[0] / 1e3, 1) if a else None,
                           "stretch": round((t / c) / (a[1] / a[0]), 2) if a else None, "share_of_kernel_time": round(t / sum(x[1] for x in sc.values()), 3)})
print(json.dumps(out, indent=1))
