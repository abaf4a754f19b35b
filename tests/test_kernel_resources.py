"""Static resources of the built gfx950 kernels (scripts/kernel_resources.py: code-object metadata, no GPU): a regression guard for what the measured
design relies on -- no kernel spills vector registers, nothing on the frame's path has a private segment (round 5: one conditionally passed flag
address had cost the LM step a byte of scratch per lane and a scratch set-up per launch), the dominant kernel's register allocation admits the five
waves per SIMD it is launched for.  profiles/r06_kernel_resources.txt is this table at the profiled library."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    lib = os.path.join(ROOT, "rgc-slam_amd", "librgc_hip.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    return m, {k["demangled"]: k for k in m.kernels_of(lib)}


def test_no_vector_spills_and_no_scratch_on_the_frame_path():
    m, ks = _table()
    assert len(ks) >= 80
    assert [n for n, k in ks.items() if k["vgpr_spill"]] == []
    with_scratch = sorted(n for n, k in ks.items() if k["scratch"])
    assert with_scratch in ([], ["k_mapreg_associate"]), with_scratch      # (f1's association keeps a small indexed array per lane: not on the odometer's path)
    for n in ("k_knn_sp<20, true, true, false>", "k_knn_sp<20, true, true, true>", "k_knn_sp<20, false, true, false>", "k_knn_coop<20, false>", "k_knn_coop<20, true>",
              "k_count<true>", "k_count<false>", "k_cells_reduce", "k_cells_scan_write<true>", "k_place", "k_rank_gather", "k_voxel_build_coop<20>", "k_voxel_patch",
              "k_lm_step", "k_linearize", "k_fitness", "k_fe_stencils", "k_fe_select", "k_deskew", "k_transform_q"):
        assert n in ks, n
        assert ks[n]["scratch"] == 0 and ks[n]["agpr"] == 0, (n, ks[n])     # no MFMA on this path: no accumulator registers either


def test_register_allocation_admits_the_launched_occupancy():
    m, ks = _table()
    assert m.waves_per_simd(ks["k_knn_sp<20, true, true, false>"]) >= 5      # the map's full search: five waves per SIMD (SpLaunch)
    assert m.waves_per_simd(ks["k_knn_sp<20, true, true, true>"]) >= 4
    assert m.waves_per_simd(ks["k_lm_step"]) >= 3 and ks["k_lm_step"]["lds"] <= 64 * 1024
    for n in ("k_count<true>", "k_place", "k_rank_gather", "k_cells_reduce", "k_transform_q"):    # the streaming passes: full occupancy
        assert m.waves_per_simd(ks[n]) == 8, n


def test_the_in_tree_library_is_what_the_sources_build_to(tmp_path):
    """A from-scratch build of the committed sources (RGC_LIB_OUT: beside the product, about a minute of hipcc) holds the same gfx950 device code as the
    in-tree librgc_hip.so the GPU tests, smoke() and bench.py load: .text of every code object by SHA-256, every kernel's registers / LDS / scratch
    (scripts/same_device_code.py).  A library left over from other sources, or from a build with RGC_EXTRA_FLAGS, fails here and not on the GPU box."""
    import subprocess
    import sys
    out = tmp_path / "librgc_rebuild.so"
    env = dict(os.environ, RGC_LIB_OUT=str(out))
    env.pop("RGC_EXTRA_FLAGS", None)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "rgc-slam_amd", "build.py"), "--force"], env=env, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "same_device_code.py"), os.path.join(ROOT, "rgc-slam_amd", "librgc_hip.so"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
