#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py tests/test_gpu_cpp_node.py -x -q --timeout 200 --timeout-method=thread 2>&1 | tail -30 > gpurun_out/s17_tests.log
tail -6 gpurun_out/s17_tests.log
for r in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s17_bench$r.json 2> gpurun_out/s17_bench.log
python - <<PY
import json
d=json.load(open('gpurun_out/s17_bench$r.json'))
print(d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'], d['timed_steps_ms']['max'])
print(d['kernel_ms_per_step'])
PY
done
timeout 300 bash scripts/prof_dependent.sh 30 0 > gpurun_out/s17_dep0_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 2000 > gpurun_out/s17_dep0_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
head -8 gpurun_out/s17_dep0_kernels.txt
