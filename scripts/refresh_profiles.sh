#!/bin/bash
# Round profile refresh on the GPU box, everything from ONE run at HEAD: bench line, rocprofv3 kernel stats of the same command, PMC passes
# and the executed instruction mix of the dominant kernel, the side benches.  Outputs under gpurun_out/${TAG}f/ ; scripts/collect_profiles.sh
# copies what should be judged into profiles/${TAG}_*.
#   usage: scripts/refresh_profiles.sh [quick]     (quick: skip the long runs and the side benches)
TAG=${RGC_ROUND_TAG:-r04}
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}f
rm -rf $O; mkdir -p $O
# 1. counters and executed instruction mix of the dominant kernel FIRST: bench.py quotes them (traffic, VALU per query, mix-weighted peak)
scripts/pmc_kernel.sh "k_knn_sp<20, true, true>" knn > /dev/null 2>&1; cp gpurun_out/pmc_knn.json $O/pmc_knn.json
scripts/pmc_kernel.sh "k_knn_sp<20, false, true>" knn_src 30000 > /dev/null 2>&1; cp gpurun_out/pmc_knn_src.json $O/pmc_knn_src.json
#    loop trip counts from a developer build (-DRGC_LAB), weights from the ISA of the product build
RGC_EXTRA_FLAGS="-DRGC_LAB" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
python3 scripts/isa_mix.py --collect > $O/lab_iters.log 2>&1; cp gpurun_out/lab_iters.json $O/lab_iters.json
python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
python3 scripts/isa_mix.py --lab $O/lab_iters.json --pmc $O/pmc_knn.json > $O/knn_isa_mix.json 2> $O/isa_mix.log
cp $O/pmc_knn.json profiles/${TAG}_pmc_knn.json; cp $O/knn_isa_mix.json profiles/${TAG}_knn_isa_mix.json   # (this box's copy of the tree: what bench.py reads)
# 2. the bench line, and the same command under rocprofv3
python bench.py > $O/bench.json 2> $O/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --configs none > $O/bench_under_rocprof.json 2> $O/rocprof.log
cd $GRAFT_REPO_ROOT
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} $O/domain_stats.csv
rm -rf $O/stats
# 3. one frame at a time under the profiler: per-kernel time of a dependent frame and its timeline
bash scripts/prof_dependent.sh 40 0 > $O/dependent_frame_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > $O/dependent_frame_timeline.txt 2>&1
if [ "$1" != quick ]; then
  python scripts/exp_long_run.py 3000 2>/dev/null | tail -1 > $O/long_run.json
  python scripts/exp_long_run_dependent.py 50 2>/dev/null | tail -1 > $O/long_run_dependent.json
  python scripts/bench_rolling.py > $O/rolling.json 2> /dev/null
  python scripts/bench_cpp_node.py > $O/cpp_node.json 2> /dev/null
  python scripts/bench_cpp_pipeline.py 2>/dev/null | tail -1 > $O/cpp_pipeline.json
  python scripts/bench_frontend.py 2>/dev/null | tail -1 > $O/frontend.json
  python scripts/bench_mapreg.py > $O/mapreg.json 2> /dev/null
  python scripts/bench_icp.py 2>/dev/null | tail -1 > $O/icp.json
  python scripts/bench_pre.py 2>/dev/null | tail -1 > $O/pre.json
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/fstats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_frontend.py > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  find $O/fstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/f_kernel_stats.csv; rm -rf $O/fstats
fi
ls -la $O; cut -c1-600 $O/bench.json
