#!/bin/bash
# Counters of the map's bulk kNN launch -- with the neighbour lists (the product), seeded without them, unseeded -- over scripts/prof_frame_reframed.py: separate rocprofv3 --pmc passes,
# per-kernel averages into gpurun_out/pmc_seeded.json.      usage: bash scripts/pmc_seeded.sh ["extra counter sets" ...]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_seeded
rm -rf $O; mkdir -p $O
SETS=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "$@")
# three sweeps: RGC_KNN_SEEDS=0 (nothing kept between targets: every launch is the FULL search, what bench.py's `value` runs), the library's
# default (neighbour lists on: launches 2..5 take certified queries' neighbours from their lists and search the rest), and RGC_KNN_CACHE=0
# (every launch searches every query, seeded: what a frame runs after the map's buffer was written to)
for mode in none lists nolists; do
  unset RGC_KNN_CACHE RGC_KNN_SEEDS
  if [ $mode = nolists ]; then export RGC_KNN_CACHE=0; fi
  if [ $mode = none ]; then export RGC_KNN_SEEDS=0; fi
  for set in "${SETS[@]}"; do
    d=$O/$mode/$(echo $set | tr ' ' '_')
    mkdir -p $O/$mode
    rocprofv3 --pmc $set -d $d -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame_reframed.py 1000000 5 > $d.log 2>&1 || tail -3 $d.log
  done
done
unset RGC_KNN_CACHE RGC_KNN_SEEDS
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections, os
root = os.environ["GRAFT_REPO_ROOT"]
O = os.path.join(root, "gpurun_out", "pmc_seeded")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for mode in ("none", "lists", "nolists"):
    for f in glob.glob(os.path.join(O, mode, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "k_knn_sp<20, true, true" in kn:
                if "true, true, true" in kn:
                    if mode != "none":
                        acc["lists" if mode == "lists" else "seeded"][r["Counter_Name"]].append(float(r["Counter_Value"]))
                elif mode == "none":
                    acc["full_search"][r["Counter_Name"]].append(float(r["Counter_Value"]))
hc = os.path.join(root, ".head_commit")
out = {"commit": open(hc).read().strip() if os.path.exists(hc) else None, "queries_per_launch": 1000000}
for which, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    o = {"per_launch": m}
    if "SQ_INSTS_VALU" in m: o["valu_wave_instructions_per_query"] = round(m["SQ_INSTS_VALU"] / 1e6, 1)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m: o["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    for a, b, name in (("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "valu_active_per_busy_cycle"), ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "wave_cycles_waiting_on_issue_frac"),
                       ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "wave_cycles_waiting_on_counters_frac"), ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "lds_per_valu"),
                       ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "vmem_rd_per_valu"), ("SQ_INST_CYCLES_VMEM", "SQ_BUSY_CYCLES", "vmem_inst_cycles_per_busy_cycle")):
        if a in m and b in m and m[b]: o[name] = round(m[a] / m[b], 4)
    out[which] = o
# the launch of the bench's timed region -- the full search of a re-framed map with nothing kept between frames (rgc_set_knn_reuse(RGC_REUSE_NONE)
# = RGC_KNN_SEEDS=0) -- at the top level, in the layout bench.py reads; beside it "lists" (the library's default on an unchanged map:
# certified queries from their neighbour lists, the rest searched) and "seeded" (the lists off: every query searched from its last k-th distance)
if "full_search" in out:
    top = dict(out.pop("full_search"))
    top["launch"] = "full_search"
    top["kernel"] = "k_knn_sp<20, true, true, false> (1000000 queries per launch; scripts/prof_frame_reframed.py under RGC_KNN_SEEDS=0: every query of every launch searched in full)"
    out = {**top, **out}
json.dump(out, open(os.path.join(root, "gpurun_out", "pmc_seeded.json"), "w"), indent=1)
print(json.dumps(out))
PY
