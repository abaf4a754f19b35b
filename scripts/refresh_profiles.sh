#!/bin/bash
# Round profile refresh on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, the side benches.
# Outputs under gpurun_out/r01f/ ; copy what should be judged into profiles/.
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r01f
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.log
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py 1000000 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py 1000000 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/make_pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json $O/pmc_traffic_raw.json > /dev/null 2>&1
python scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
python scripts/bench_rolling.py > $O/rolling.json 2> /dev/null
python scripts/bench_cpp_node.py > $O/cpp_node.json 2> /dev/null
python scripts/bench_frontend.py 2>/dev/null | tail -1 > $O/frontend.json
python scripts/bench_mapreg.py > $O/mapreg.json 2> /dev/null
python scripts/bench_icp.py 2>/dev/null | tail -1 > $O/icp.json
python scripts/bench_pre.py 2>/dev/null | tail -1 > $O/pre.json
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} $O/domain_stats.csv
rm -rf $O/stats $O/fetch $O/write
ls -la $O; cat $O/bench.json | cut -c1-400
