#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lazy or edge or golden or oracle_parity" --timeout 300 --timeout-method=thread 2>&1 | tail -8 > gpurun_out/s11_tests.log
tail -4 gpurun_out/s11_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --configs c3,c5 --no-cpu-baseline > gpurun_out/s11_bench.json 2> gpurun_out/s11_bench.log
cut -c1-200 gpurun_out/s11_bench.json
timeout 300 bash scripts/prof_dependent.sh 30 1 cmain 2 > gpurun_out/s11_lazy_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s11_lazy_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
python - <<'PY'
import json
d=json.load(open('gpurun_out/s11_bench.json'))
print(d['value'], d['lazy_target']['two_contexts'], d['lazy_target']['one_frame_at_a_time'])
for c in d['configs']: print({k:v for k,v in c.items() if k in ('config','scans_per_s','lazy_target_scans_per_s','lazy_target_same_poses')})
PY
head -8 gpurun_out/s11_lazy_kernels.txt
