#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sequence.py tests/test_gpu_cpp_node.py tests/test_gpu_rolling_map.py -x -q --timeout 300 --timeout-method=thread 2>&1 | tail -30 > gpurun_out/s13_tests.log
tail -6 gpurun_out/s13_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s13_bench.json 2> gpurun_out/s13_bench.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/s13_bench.json'))
print(d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['steady_state'].items() if isinstance(v,dict)}, d['lazy_target']['two_contexts']['ms_per_step'])
print(d['kernel_ms_per_step'])
PY
timeout 300 bash scripts/prof_dependent.sh 30 0 > gpurun_out/s13_dep0_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s13_dep0_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
head -12 gpurun_out/s13_dep0_kernels.txt
timeout 300 python scripts/bench_rolling.py > gpurun_out/s13_rolling.json 2>/dev/null; cut -c1-700 gpurun_out/s13_rolling.json
