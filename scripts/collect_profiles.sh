#!/bin/bash
# Copy the outputs of scripts/refresh_profiles.sh (gpurun_out/${TAG}f/) into profiles/${TAG}_* and regenerate the figures the documents quote.
TAG=${RGC_ROUND_TAG:-r06}
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/${TAG}f
for f in bench.json bench_under_rocprof.json kernel_stats.csv domain_stats.csv dependent_frame_kernels.txt dependent_frame_timeline.txt dependent_two_contexts_kernels.txt dependent_two_contexts_timeline.txt lazy_target_kernels.txt lazy_target_timeline.txt pmc_knn.json pmc_knn_src.json lab_iters.json lab_seeded.jsonl knn_isa_mix.json knn_isa_mix_seeded.json frame_traffic.json frame_traffic_lists.json frame_kernel_times.json exp_sequences.jsonl seq_concurrency_S4.json frame_traffic_c3.json frame_traffic_c5.json long_run.json long_run_dependent.json general_route.json; do
  [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f
done
for p in rolling:rolling_bench cpp_node:cpp_node_bench cpp_pipeline:cpp_pipeline_bench frontend:frontend_bench mapreg:mapreg_bench icp:icp_bench pre:pre_bench; do
  [ -s $S/${p%%:*}.json ] && cp $S/${p%%:*}.json profiles/${TAG}_${p##*:}.json
done
[ -s $S/f_kernel_stats.csv ] && cp $S/f_kernel_stats.csv profiles/${TAG}_frontend_kernel_stats.csv
python3 scripts/sync_docs.py
