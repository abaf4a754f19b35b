#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for lib in prev cur; do
if [ $lib = prev ]; then export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_build/librgc_prev.so; else unset RGC_HIP_LIB; fi
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s18_$lib$r.json 2> gpurun_out/s18_bench.log
python - <<PY
import json
d=json.load(open('gpurun_out/s18_$lib$r.json'))
print("$lib", d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'], d['lazy_target']['one_frame_at_a_time']['ms_per_step'], d['timed_steps_ms']['max'])
PY
done
done
unset RGC_HIP_LIB
timeout 300 python scripts/bench_rolling.py > gpurun_out/s18_rolling.json 2>/dev/null; cut -c1-420 gpurun_out/s18_rolling.json
