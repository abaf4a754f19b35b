"""The short sequence tests/test_gpu_parity.py::test_environment_knobs_change_no_result runs, for scripts/exp_build_flags.sh."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import registration
    world, tgt = synth.make_world_and_map(100000, seed=synth.SEED)
    poses = synth.make_trajectory(4, seed=synth.SEED + 11)
    scans = [synth.make_scan_n(world, poses[i + 1], 12000, seed=synth.SEED + 900 + i)["xyz"] for i in range(3)]
    v = registration.odometer_vgicp(0)
    g, out = poses[0].astype(np.float32), []
    for s_ in scans:
        v.setInputTarget(tgt)
        v.setInputSource(s_)
        v.align(g, want_output=False, want_fitness=True)
        g = v.getFinalTransformation()
        out.append((g, int(v.nr_iterations), float(v.getFitnessScore())))
    v.close()
    return out
