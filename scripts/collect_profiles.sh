#!/bin/bash
# Copy the outputs of scripts/refresh_profiles.sh (gpurun_out/r02f/) into profiles/r02_* and regenerate the figures the documents quote.
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/r02f
for f in bench.json bench_under_rocprof.json kernel_stats.csv domain_stats.csv pmc_knn.json pmc_knn_src.json valu_issue.jsonl stall.txt long_run.json; do cp $S/$f profiles/r02_$f; done
[ -s $S/long_run_pipelined.json ] && cp $S/long_run_pipelined.json profiles/r02_long_run_pipelined.json
cp $S/rolling.json profiles/r02_rolling_bench.json; cp $S/cpp_node.json profiles/r02_cpp_node_bench.json; cp $S/cpp_pipeline.json profiles/r02_cpp_pipeline_bench.json
cp $S/frontend.json profiles/r02_frontend_bench.json; cp $S/mapreg.json profiles/r02_mapreg_bench.json; cp $S/icp.json profiles/r02_icp_bench.json; cp $S/pre.json profiles/r02_pre_bench.json
cp $S/f_kernel_stats.csv profiles/r02_frontend_kernel_stats.csv; cp $S/m_kernel_stats.csv profiles/r02_mapreg_kernel_stats.csv; cp $S/i_kernel_stats.csv profiles/r02_icp_kernel_stats.csv
python3 scripts/sync_docs.py
