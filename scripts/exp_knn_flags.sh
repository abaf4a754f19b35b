#!/bin/bash
# Build-flag variants of the map's kNN kernel (RGC_SPBUF/RGC_SPLOW/RGC_KNN_T/RGC_XCD_RUN), built beforehand into exp_flags/librgc_<name>.so
# (RGC_EXTRA_FLAGS=... RGC_LIB_OUT=... python3 rgc-slam_amd/build.py), each timed on the dependent c-main sequence with nothing kept
# between frames (scripts/exp_runtime.py): frames per second + a checksum of the poses (must not move).
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/exp_knn_flags.jsonl
for name in base "$@" base; do
  lib=$PWD/exp_flags/librgc_$name.so
  [ -s $lib ] || { echo "no $lib"; continue; }
  echo -n "{\"variant\": \"$name\", \"run\": " >> gpurun_out/exp_knn_flags.jsonl
  RGC_HIP_LIB=$lib timeout 300 python3 scripts/exp_runtime.py system 40 2>/dev/null | tail -1 | tr -d '\n' >> gpurun_out/exp_knn_flags.jsonl
  echo "}" >> gpurun_out/exp_knn_flags.jsonl
done
cat gpurun_out/exp_knn_flags.jsonl
