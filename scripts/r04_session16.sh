#!/bin/bash
cd $GRAFT_REPO_ROOT
export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_build/librgc_exp.so
for e in 0 1 2 0 1 2; do
RGC_EXP=$e timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s16_bench$e.json 2> gpurun_out/s16_bench.log
python - <<PY
import json
d=json.load(open('gpurun_out/s16_bench$e.json'))
print($e, d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'], d['timed_steps_ms']['max'])
PY
done
for e in 1 2; do
RGC_EXP=$e timeout 250 bash scripts/prof_dependent_api.sh 12 0 > gpurun_out/s16_api_exp$e.txt 2>&1
done
rm -rf gpurun_out/prof_dep_api
