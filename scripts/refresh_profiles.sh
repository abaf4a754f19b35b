#!/bin/bash
# Round profile refresh on the GPU box, everything from ONE run at HEAD: bench line, rocprofv3 kernel stats of the same command, PMC passes
# and the executed instruction mix of the dominant kernel, frame-level HBM traffic (c-main, c3, c5), the side benches.  Outputs under
# gpurun_out/${TAG}f/ ; scripts/collect_profiles.sh copies what should be judged into profiles/${TAG}_*.  Every step runs under `timeout`.
#   usage: scripts/refresh_profiles.sh [quick]     (quick: skip the long runs, c3 / c5 traffic and the side benches)
TAG=${RGC_ROUND_TAG:-r06}
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}f
rm -rf $O; mkdir -p $O
# 1. counters and executed instruction mix of the dominant kernel FIRST: bench.py quotes them (traffic, VALU per query, mix-weighted peak)
#    (round 6: bench.py's `value` keeps nothing between frames -- the dominant launch is the FULL search of the re-framed map, at the top level of
#    pmc_knn.json; the library's default on an unchanged map ("lists") and the seeded search ("seeded") are in the same file)
timeout 1800 bash scripts/pmc_seeded.sh "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" > /dev/null 2>&1; cp gpurun_out/pmc_seeded.json $O/pmc_knn.json
timeout 900 scripts/pmc_kernel.sh "k_knn_sp<20, false, true" knn_src 30000 > /dev/null 2>&1; cp gpurun_out/pmc_knn_src.json $O/pmc_knn_src.json
#    loop trip counts from a developer build (-DRGC_LAB) beside the product, weights from the ISA of the product build
RGC_EXTRA_FLAGS="-DRGC_LAB" RGC_LIB_OUT=/tmp/librgc_lab.so timeout 600 python3 rgc-slam_amd/build.py > /dev/null 2>&1
#    (the loop counts and the mix describe the SEARCHES -- seeded with the lists off, and unseeded; the launch with the lists on is
#    straight-line code per certified query: its counters are in pmc_knn.json's top level)
RGC_KNN_CACHE=0 RGC_HIP_LIB=/tmp/librgc_lab.so timeout 300 python3 scripts/isa_mix.py --collect > $O/lab_iters.log 2>&1; cp gpurun_out/lab_iters.json $O/lab_iters.json
python3 -c "import json; d=json.load(open('$O/pmc_knn.json')); json.dump(d['seeded'], open('$O/pmc_knn_seeded.json','w'), indent=1); json.dump({k: v for k, v in d.items() if k not in ('seeded', 'lists')}, open('$O/pmc_knn_full.json','w'), indent=1)"
timeout 600 python3 scripts/isa_mix.py --lab $O/lab_iters.json --pmc $O/pmc_knn_full.json > $O/knn_isa_mix.json 2> $O/isa_mix.log
timeout 600 python3 scripts/isa_mix.py --seeded --lab $O/lab_iters.json --pmc $O/pmc_knn_seeded.json > $O/knn_isa_mix_seeded.json 2>> $O/isa_mix.log
RGC_HIP_LIB=/tmp/librgc_lab.so timeout 300 python3 scripts/lab_seeded.py 1000000 4 > $O/lab_seeded.jsonl 2>&1
# 2. frame-level traffic, measured (every kernel of a dependent frame, nothing kept between frames like `value`): c-main always, c3 / c5 in the
#    full run; the library's default on the unchanged map beside it (lists); GPU time per kernel and frame of the timed workload
timeout 600 bash scripts/frame_traffic.sh 10 cmain > $O/frame_traffic_cmain.log 2>&1; cp gpurun_out/frame_traffic_cmain.json $O/frame_traffic.json
timeout 600 bash scripts/frame_traffic.sh 10 cmain lists > $O/frame_traffic_cmain_lists.log 2>&1; cp gpurun_out/frame_traffic_cmain_lists.json $O/frame_traffic_lists.json
timeout 600 bash scripts/frame_kernel_times.sh 40 1 cmain > $O/frame_kernel_times.log 2>&1; cp gpurun_out/frame_kernel_times.json $O/frame_kernel_times.json
cp $O/pmc_knn.json profiles/${TAG}_pmc_knn.json; cp $O/knn_isa_mix.json profiles/${TAG}_knn_isa_mix.json; cp $O/frame_traffic.json profiles/${TAG}_frame_traffic.json; cp $O/frame_kernel_times.json profiles/${TAG}_frame_kernel_times.json   # (this box's copy of the tree: what bench.py reads)
# 3. the bench line, and the same command under rocprofv3
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --configs none > $O/bench_under_rocprof.json 2> $O/rocprof.log
cd $GRAFT_REPO_ROOT
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} $O/domain_stats.csv
rm -rf $O/stats
# 4. timelines under the profiler: a dependent frame one at a time, on two contexts, and with the lazy target -- nothing kept between frames
export RGC_KNN_SEEDS=0
timeout 400 bash scripts/prof_dependent.sh 40 0 > $O/dependent_frame_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > $O/dependent_frame_timeline.txt 2>&1
timeout 400 bash scripts/prof_dependent.sh 40 1 > $O/dependent_two_contexts_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > $O/dependent_two_contexts_timeline.txt 2>&1
timeout 400 bash scripts/prof_dependent.sh 40 1 cmain 2 > $O/lazy_target_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > $O/lazy_target_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
unset RGC_KNN_SEEDS
# 5. what one GPU carries: S sequences from C++ host threads, and the kernel trace of S = 4
timeout 900 bash scripts/exp_sequences.sh > $O/exp_sequences.log 2>&1; cp gpurun_out/exp_sequences.jsonl $O/exp_sequences.jsonl; cp gpurun_out/seq_concurrency_S4.json $O/seq_concurrency_S4.json
if [ "$1" != quick ]; then
  timeout 900 bash scripts/frame_traffic.sh 3 c3 > $O/frame_traffic_c3.log 2>&1; cp gpurun_out/frame_traffic_c3.json $O/frame_traffic_c3.json
  timeout 1200 bash scripts/frame_traffic.sh 2 c5 > $O/frame_traffic_c5.log 2>&1; cp gpurun_out/frame_traffic_c5.json $O/frame_traffic_c5.json
  timeout 600 python scripts/exp_long_run.py 3000 2>/dev/null | tail -1 > $O/long_run.json
  timeout 600 python scripts/exp_long_run_dependent.py 50 2>/dev/null | tail -1 > $O/long_run_dependent.json
  timeout 600 python scripts/bench_rolling.py > $O/rolling.json 2> /dev/null
  timeout 900 python scripts/bench_cpp_node.py > $O/cpp_node.json 2> /dev/null
  timeout 600 python scripts/bench_cpp_pipeline.py 2>/dev/null | tail -1 > $O/cpp_pipeline.json
  timeout 300 python scripts/bench_frontend.py 2>/dev/null | tail -1 > $O/frontend.json
  timeout 300 python scripts/bench_mapreg.py > $O/mapreg.json 2> /dev/null
  timeout 300 python scripts/bench_icp.py 2>/dev/null | tail -1 > $O/icp.json
  timeout 300 python scripts/bench_pre.py 2>/dev/null | tail -1 > $O/pre.json
  timeout 300 python scripts/bench_general.py > $O/general_route.json 2> /dev/null
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/fstats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_frontend.py > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  find $O/fstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/f_kernel_stats.csv; rm -rf $O/fstats
fi
ls -la $O; cut -c1-600 $O/bench.json
