#!/bin/bash
# A/B of two builds on the bench's stage times: bash scripts/r04_ab.sh "<flags A>" "<flags B>"
cd $GRAFT_REPO_ROOT
i=0
for fl in "$1" "$2"; do
  i=$((i+1))
  out=/tmp/librgc_ab$i.so
  RGC_EXTRA_FLAGS="$fl" RGC_LIB_OUT=$out python rgc-slam_amd/build.py > /dev/null 2>&1
  RGC_HIP_LIB=$out timeout 400 python bench.py --steps 12 --warmup 3 --configs none --no-cpu-baseline > gpurun_out/ab$i.json 2> gpurun_out/ab$i.log
  python - <<PY
import json
d=json.load(open("gpurun_out/ab$i.json"))
print("flags [$fl]:", d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["steady_state"].items() if isinstance(v,dict)}, d["kernel_ms_per_step"], "lazy", d["lazy_target"]["two_contexts"]["ms_per_step"])
PY
done
