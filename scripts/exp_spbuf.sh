#!/bin/bash
# Developer aid: the append buffer's depth and the level a full buffer is drained to (RGC_SPBUF / RGC_SPLOW) against the chain inserts a
# wave pays and the launch time (run on the GPU box).   bash scripts/exp_spbuf.sh "12:0 12:6 12:8 16:10"
cd "$GRAFT_REPO_ROOT"
for v in ${1:-12:0 12:4 12:6 12:7 12:8 16:8 16:10 16:12}; do
  n=${v%%:*}; lo=${v##*:}
  RGC_EXTRA_FLAGS="-DRGC_LAB -DRGC_SPBUF=$n -DRGC_SPLOW=$lo" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  it=$(python3 scripts/isa_mix.py --collect 2>/dev/null | tail -1)
  RGC_EXTRA_FLAGS="-DRGC_SPBUF=$n -DRGC_SPLOW=$lo" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  t=$(python3 scripts/lab_knn.py 1000000 10 2>/dev/null | tail -1)
  echo "{\"spbuf\": $n, \"splow\": $lo, \"lab\": $it, \"timing\": $t}"
done | tee gpurun_out/exp_spbuf.jsonl
python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
