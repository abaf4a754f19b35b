"""The C++ host mirror at the headline size: rgc::PipelinedVGICP (two contexts taking turns) against rgc::FastVGICPHip one frame at a time,
30 k-point scans vs a 1 M-point map rebuilt every frame, clouds resident on the device (tests/cpp/test_pipelined.cpp)."""
import sys, os, json, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rgc_slam_amd.synth as synth
n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, "test_pipelined")
    subprocess.check_call(["g++", "-std=c++14", "-O2", os.path.join(ROOT, "tests", "cpp", "test_pipelined.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "rgc-slam_amd"), "-lrgc_hip", "-Wl,-rpath," + os.path.join(ROOT, "rgc-slam_amd")])
    world, tgt = synth.make_world_and_map(n_t)
    poses = synth.make_trajectory(9)
    scans = [synth.make_scan_n(world, poses[i + 1], 30000, seed=synth.SEED + 100 + i)["xyz"] for i in range(8)]
    def dump(a, path):
        with open(path, "wb") as f:
            a = np.ascontiguousarray(a, dtype=np.float32)
            f.write(np.int32(len(a)).tobytes()); f.write(a.tobytes())
    dump(tgt, os.path.join(d, "t.bin"))
    for i, s in enumerate(scans):
        dump(s, os.path.join(d, f"s{i}.bin"))
    out = subprocess.run([exe, os.path.join(d, "t.bin"), "8"] + [os.path.join(d, f"s{i}.bin") for i in range(8)] + ["12"],
                         capture_output=True, text=True, timeout=900).stdout
lines = dict(l.split(" ", 1) for l in out.strip().splitlines())
a, b = float(lines["ms_per_frame_one_at_a_time"]), float(lines["ms_per_frame_pipelined"])
print(json.dumps({"workload": f"C++ host (rgc::PipelinedVGICP): 30000-pt scans vs a {n_t}-pt map rebuilt every frame, device-resident clouds, {lines['frames']} frames",
                  "one_at_a_time_ms_per_frame": a, "one_at_a_time_scans_per_s": round(1e3 / a, 1), "pipelined_ms_per_frame": b,
                  "pipelined_scans_per_s": round(1e3 / b, 1), "same_poses_and_fitness": lines["same"] == "1"}))
