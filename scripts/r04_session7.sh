#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_rolling_map.py tests/test_gpu_sequence.py -x -q 2>&1 | tail -25 > gpurun_out/s7_tests.log
python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s7_bench.json 2> gpurun_out/s7_bench.log
bash scripts/prof_dependent.sh 30 1 > gpurun_out/s7_dep1_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s7_dep1_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
tail -4 gpurun_out/s7_tests.log; cut -c1-200 gpurun_out/s7_bench.json
