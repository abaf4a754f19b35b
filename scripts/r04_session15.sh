#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py -x -q --timeout 300 --timeout-method=thread 2>&1 | tail -5 > gpurun_out/s15_tests.log
tail -3 gpurun_out/s15_tests.log
for r in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s15_bench$r.json 2> gpurun_out/s15_bench.log
python - <<PY
import json
d=json.load(open('gpurun_out/s15_bench$r.json'))
print(d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'], d['timed_steps_ms']['max'])
PY
done
timeout 250 bash scripts/prof_dependent_api.sh 12 0 > gpurun_out/s15_api0.txt 2>&1
rm -rf gpurun_out/prof_dep_api
