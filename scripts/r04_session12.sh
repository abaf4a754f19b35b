#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  RGC_TRACE_ALLOC=1 timeout 300 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline > gpurun_out/s12_b$i.json 2> gpurun_out/s12_b$i.log
  python - <<PY
import json
d=json.load(open("gpurun_out/s12_b$i.json"))
print($i, d["value"], d["ms_per_step"], d["timed_steps_ms"]["median"], d["timed_steps_ms"]["max"], d["timed_steps_ms"]["slowest_step"], d["one_frame_at_a_time"]["ms_per_step"], d["lazy_target"]["two_contexts"]["ms_per_step"])
PY
  grep -c "grew\|left its" gpurun_out/s12_b$i.log
done
timeout 600 python scripts/prof_cpp_node.py > gpurun_out/s12_cpp_node_prof.txt 2>&1
tail -60 gpurun_out/s12_cpp_node_prof.txt
