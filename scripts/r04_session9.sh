#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py > gpurun_out/s9_bench.json 2> gpurun_out/s9_bench.log
timeout 900 bash scripts/frame_traffic.sh 3 c3 > gpurun_out/s9_ft_c3.log 2>&1
timeout 1200 bash scripts/frame_traffic.sh 2 c5 > gpurun_out/s9_ft_c5.log 2>&1
timeout 300 python scripts/bench_frontend.py > gpurun_out/s9_frontend.json 2> gpurun_out/s9_frontend.log
cut -c1-300 gpurun_out/s9_bench.json; tail -3 gpurun_out/s9_ft_c3.log | cut -c1-300; tail -3 gpurun_out/s9_ft_c5.log | cut -c1-300; cat gpurun_out/s9_frontend.json
