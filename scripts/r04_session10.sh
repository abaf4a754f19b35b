#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_rolling_map.py tests/test_gpu_sequence.py -x -q --timeout 300 --timeout-method=thread 2>&1 | tail -30 > gpurun_out/s10_tests.log
tail -6 gpurun_out/s10_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --configs c3,c5 --no-cpu-baseline > gpurun_out/s10_bench.json 2> gpurun_out/s10_bench.log
cut -c1-200 gpurun_out/s10_bench.json
timeout 300 bash scripts/prof_dependent.sh 30 1 cmain 2 > gpurun_out/s10_lazy_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s10_lazy_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
