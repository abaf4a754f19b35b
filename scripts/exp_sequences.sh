#!/bin/bash
# S independent sequences on one GPU from C++ host threads (rgc-slam_amd/cpp/sequences_per_gpu.cpp) under different numbers of hardware
# queues (GPU_MAX_HW_QUEUES: the HIP runtime maps a process's streams onto that many; default 4 -- S sequences are 4 S streams), and
# with the library's default reuse mode.   usage (GPU box): bash scripts/exp_sequences.sh   -> gpurun_out/exp_sequences.jsonl
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp_sequences.jsonl; : > $O
D=/tmp/rgc_seq_data; rm -rf $D; mkdir -p $D
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import rgc_slam_amd.synth as synth
F = 24
for k in range(2):
    seed = synth.SEED + k
    world, tgt = synth.make_world_and_map(1000000, seed=seed)
    poses = synth.make_trajectory(F + 1, seed=seed)
    d = f"/tmp/rgc_seq_data/d{k}"; os.makedirs(d)
    def dump(a, path):
        a = np.ascontiguousarray(a, np.float32)
        with open(path, "wb") as f: f.write(np.int32(len(a)).tobytes()); f.write(a.tobytes())
    dump(tgt, d + "/map.bin")
    open(d + "/pose0.bin", "wb").write(np.ascontiguousarray(poses[0], np.float64).tobytes())
    for i in range(F):
        dump(synth.make_scan_n(world, poses[i + 1], 30000, seed=seed + 100 + i)["xyz"], d + f"/s{i}.bin")
PY
g++ -std=c++14 -O2 -pthread rgc-slam_amd/cpp/sequences_per_gpu.cpp -o /tmp/seqs -L rgc-slam_amd -lrgc_hip -Wl,-rpath,$GRAFT_REPO_ROOT/rgc-slam_amd
for q in default 8 16 32; do
  for reuse in 0 2; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    echo "{\"GPU_MAX_HW_QUEUES\": \"$q\", \"reuse\": $reuse, \"result\": $(timeout 300 /tmp/seqs 0 24 6 $reuse 1,2,4,8 2 $D/d0 $D/d1)}" >> $O
  done
done
unset GPU_MAX_HW_QUEUES
cat $O
# the kernel trace of S = 4 (every sequence alone first, then all four at once): scripts/seq_concurrency.py says what stretches
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/seq_prof; timeout 300 rocprofv3 --kernel-trace -d /tmp/seq_prof -o s4 --output-format csv -- /tmp/seqs 0 24 8 0 4 2 $D/d0 $D/d1 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python3 scripts/seq_concurrency.py /tmp/seq_prof 4 0.55 > gpurun_out/seq_concurrency_S4.json 2>&1; cat gpurun_out/seq_concurrency_S4.json
