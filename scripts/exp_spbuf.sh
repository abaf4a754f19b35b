#!/bin/bash
# Developer aid: the append buffer's depth against the chain inserts a wave pays and the launch time (run on the GPU box).
cd "$GRAFT_REPO_ROOT"
for n in 12 16 20 24 32; do
  RGC_EXTRA_FLAGS="-DRGC_LAB -DRGC_SPBUF=$n" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  it=$(python3 scripts/isa_mix.py --collect 2>/dev/null | tail -1)
  RGC_EXTRA_FLAGS="-DRGC_SPBUF=$n" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  t=$(python3 scripts/lab_knn.py 1000000 10 2>/dev/null | tail -1)
  echo "{\"spbuf\": $n, \"lab\": $it, \"timing\": $t}"
done | tee gpurun_out/exp_spbuf.jsonl
python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
