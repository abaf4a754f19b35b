#!/bin/bash
# Instruction-issue counters of the dominant kernel (k_knn_rows<20,true>, the map's bulk kNN launch): separate rocprofv3 --pmc
# passes over scripts/prof_frame.py (30 k-point scan vs 1 M-point map), summarised into gpurun_out/pmc_issue.json
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_issue
rm -rf $O; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  d=$O/$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set -d $d -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py 1000000 4 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_issue")
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_knn_rows<20, true>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
out = {"kernel": "k_knn_rows<20, true> (1 M queries per launch)", "per_launch": m}
q = 1.0e6
if "SQ_INSTS_VALU" in m:
    out["valu_wave_instructions_per_query"] = round(m["SQ_INSTS_VALU"] / q, 1)
if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m:
    out["note"] = "SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES etc. are summed over shader engines as rocprofv3 reports them; ratios within one pass are meaningful"
for a, b, name in (("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "valu_active_per_busy_cycle"), ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "wave_cycles_waiting_frac"),
                   ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "lds_per_valu"), ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "vmem_rd_per_valu"), ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "salu_per_valu")):
    if a in m and b in m and m[b]:
        out[name] = round(m[a] / m[b], 4)
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_issue.json"), "w"), indent=1)
print(json.dumps(out))
PY
