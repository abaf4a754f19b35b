#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s6_tests.log
python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.log
bash scripts/prof_dependent.sh 30 0 > gpurun_out/s6_dep0_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s6_dep0_timeline.txt 2>&1
bash scripts/prof_dependent.sh 30 1 > gpurun_out/s6_dep1_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s6_dep1_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
tail -3 gpurun_out/s6_tests.log; cut -c1-300 gpurun_out/s6_bench.json
