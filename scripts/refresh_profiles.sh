#!/bin/bash
# Round profile refresh on the GPU box, everything from ONE run at HEAD: bench line, rocprofv3 kernel stats of the same command, PMC passes
# of the dominant kernel, the side benches.  Outputs under gpurun_out/r02f/ ; copy what should be judged into profiles/ (r02_*).
#   usage: scripts/refresh_profiles.sh [quick]     (quick: skip c5, the 3000-frame run and the side benches)
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r02f
rm -rf $O; mkdir -p $O
python bench.py --configs c1,c3$([ "$1" = quick ] || echo ,c5) > $O/bench.json 2> $O/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --configs none > $O/bench_under_rocprof.json 2> $O/rocprof.log
cd $GRAFT_REPO_ROOT
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} $O/domain_stats.csv
rm -rf $O/stats
scripts/pmc_kernel.sh "k_knn_sp<20, true>" knn > /dev/null 2>&1; cp gpurun_out/pmc_knn.json $O/pmc_knn.json
scripts/pmc_kernel.sh "k_knn_sp<20, false>" knn_src 30000 > /dev/null 2>&1; cp gpurun_out/pmc_knn_src.json $O/pmc_knn_src.json
timeout 120 scripts/ubench/valu_issue > $O/valu_issue.jsonl 2> /dev/null
python scripts/exp_stall.py 300 > $O/stall.txt 2>&1; EXP_GC=freeze python scripts/exp_stall.py 300 >> $O/stall.txt 2>&1
if [ "$1" != quick ]; then
  python scripts/exp_long_run.py 3000 2>/dev/null | tail -1 > $O/long_run.json
  python scripts/exp_long_run_pipelined.py 6000 2>/dev/null | tail -1 > $O/long_run_pipelined.json
  python scripts/bench_rolling.py > $O/rolling.json 2> /dev/null
  python scripts/bench_cpp_node.py > $O/cpp_node.json 2> /dev/null
  python scripts/bench_cpp_pipeline.py 2>/dev/null | tail -1 > $O/cpp_pipeline.json
  python scripts/bench_frontend.py 2>/dev/null | tail -1 > $O/frontend.json
  python scripts/bench_mapreg.py > $O/mapreg.json 2> /dev/null
  python scripts/bench_icp.py 2>/dev/null | tail -1 > $O/icp.json
  python scripts/bench_pre.py 2>/dev/null | tail -1 > $O/pre.json
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/fstats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_frontend.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d $O/mstats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_mapreg.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d $O/istats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_icp.py > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  for t in f m i; do find $O/${t}stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${t}_kernel_stats.csv; rm -rf $O/${t}stats; done
fi
ls -la $O; cut -c1-600 $O/bench.json
