#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 60 python scripts/repro_edge.py > gpurun_out/dbg_new.log 2>&1; echo "rc=$?" >> gpurun_out/dbg_new.log
cat gpurun_out/dbg_new.log
timeout 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_rolling_map.py tests/test_gpu_sequence.py tests/test_gpu_cpp_node.py -x -q --timeout 300 --timeout-method=thread 2>&1 | tail -30 > gpurun_out/s8_tests.log
tail -12 gpurun_out/s8_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s8_bench.json 2> gpurun_out/s8_bench.log
cut -c1-200 gpurun_out/s8_bench.json
timeout 300 bash scripts/prof_dependent.sh 30 1 > gpurun_out/s8_dep1_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s8_dep1_timeline.txt 2>&1
timeout 300 bash scripts/prof_dependent.sh 30 1 cmain 2 > gpurun_out/s8_lazy_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s8_lazy_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
