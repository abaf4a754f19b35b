#!/bin/bash
# A/B of prebuilt libraries on the bench's keys (same box, alternating): bash scripts/r04_ab_libs.sh <rounds> <lib.so|cur> ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for lib in "$@"; do
if [ $lib = cur ]; then unset RGC_HIP_LIB; else export RGC_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/ab_run.json 2> gpurun_out/ab_run.log
python - <<PY
import json
d=json.load(open('gpurun_out/ab_run.json'))
print("$lib", "value", d['value'], "one-frame", d['one_frame_at_a_time']['ms_per_step'], "steady", {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, "lazy", d['lazy_target']['two_contexts']['ms_per_step'], "replay", d['replay_of_preframed_maps']['scans_per_s'], "lin", d['kernel_ms_per_step']['linearize'])
PY
done
done
