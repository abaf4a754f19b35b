#!/bin/bash
# Instruction-issue / memory counters of ONE kernel (name substring $1, e.g. "k_knn_sp<20, true>"): separate rocprofv3 --pmc passes over
# scripts/prof_frame.py (30 k-point scan vs 1 M-point map), summarised into gpurun_out/pmc_$2.json.   usage: pmc_kernel.sh <pattern> <tag> [queries per launch, default 1000000]
PAT="$1"; TAG="${2:-kernel}"; NQ="${3:-1000000}"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set -d $d -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py 1000000 4 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
PAT="$PAT" TAG="$TAG" NQ="$NQ" python3 - <<'PY'
import csv, glob, json, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_" + os.environ["TAG"])
pat = os.environ["PAT"]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
q = float(os.environ["NQ"])
hc = os.path.join(os.environ["GRAFT_REPO_ROOT"], ".head_commit")   # written beside the snapshot before the GPU call (there is no .git on the box)
out = {"kernel": pat + " (%d queries per launch)" % int(q), "commit": open(hc).read().strip() if os.path.exists(hc) else None, "per_launch": m}
if "SQ_INSTS_VALU" in m: out["valu_wave_instructions_per_query"] = round(m["SQ_INSTS_VALU"] / q, 1)
if "FETCH_SIZE" in m and "WRITE_SIZE" in m: out["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
for a, b, name in (("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "valu_active_per_busy_cycle"), ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "wave_cycles_waiting_on_issue_frac"),
                   ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "wave_cycles_waiting_on_counters_frac"), ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "lds_per_valu"),
                   ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "vmem_rd_per_valu"), ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "salu_per_valu")):
    if a in m and b in m and m[b]: out[name] = round(m[a] / m[b], 4)
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_" + os.environ["TAG"] + ".json"), "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O
