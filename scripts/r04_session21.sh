#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py -x -q --timeout 120 --timeout-method=thread 2>&1 | tail -3
for r in 1 2 3; do
for lib in prev cur; do
if [ $lib = prev ]; then export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_build/librgc_prev.so; else unset RGC_HIP_LIB; fi
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s21_$lib$r.json 2> gpurun_out/s21_bench.log
python - <<PY
import json
d=json.load(open('gpurun_out/s21_$lib$r.json'))
print("$lib", d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'], d['lazy_target']['one_frame_at_a_time']['ms_per_step'], d['kernel_ms_per_step']['linearize'])
PY
done
done
