#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/lab_knn.py 1000000 5 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method=thread 2>&1 | tail -3
for r in 1; do
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s20_bench$r.json 2> gpurun_out/s20_bench.log
python - <<PY
import json
d=json.load(open('gpurun_out/s20_bench$r.json'))
print(d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'], d['lazy_target']['one_frame_at_a_time']['ms_per_step'], d['timed_steps_ms']['max'])
print(d['kernel_ms_per_step'])
PY
done
timeout 300 python scripts/bench_rolling.py > gpurun_out/s20_rolling.json 2>/dev/null; cut -c1-420 gpurun_out/s20_rolling.json
