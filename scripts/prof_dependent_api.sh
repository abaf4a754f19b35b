#!/bin/bash
# Developer aid (GPU box): the host's HIP calls and the kernels of a few dependent frames on ONE time axis -- what the host does between
# the solve's posted result and the next frame's first kernel.   bash scripts/prof_dependent_api.sh [frames] [overlap 0|1] [cfg] [lazy margin]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_dep_api
rm -rf $O; mkdir -p $O
rocprofv3 --hip-runtime-trace --kernel-trace -d $O -o dep --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_dependent.py ${1:-12} ${2:-0} ${3:-cmain} ${4:-0} > $O/run.log 2>&1
python3 - <<PY
import csv, glob
api = list(csv.DictReader(open(glob.glob("$O/**/*hip_api_trace.csv", recursive=True)[0])))
ker = list(csv.DictReader(open(glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0])))
ev = []
for r in api:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host", r["Function"], r.get("Thread_Id", "")))
for r in ker:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gpu ", r["Kernel_Name"].replace("rgck::", "").split("(")[0][:40], "q" + r.get("Queue_Id", "")))
ev.sort()
# the last-but-three k_count<true>: print everything from 60 us before it
starts = [e[0] for e in ev if e[2] == "gpu " and e[3].startswith("void k_count<true>") or e[3].startswith("k_count<true>")]
if not starts:
    starts = [e[0] for e in ev if e[2] == "gpu " and "k_count<true>" in e[3]]
for t_ref in starts[-4:-2]:
    print("---- around the k_count<true> at", t_ref)
    for s, e, kind, name, extra in ev:
        if t_ref - 70000 <= s <= t_ref + 25000:
            print("  %8.1f us  +%7.1f  %s  %-44s %s" % ((s - t_ref) / 1e3, (e - s) / 1e3, kind, name, extra))
PY
