"""Regenerate the number-bearing blocks of DESIGN.md (section 9), BASELINE.md (section 4), README.md and profiles/README.md from the files
under profiles/r06_* (one refresh = one run of scripts/refresh_profiles.sh + scripts/collect_profiles.sh), so that the prose cannot drift
from the measurements.  A block lives between `<!-- r06-numbers:begin ... -->` and `<!-- r06-numbers:end -->`; everything else in those
documents is written by hand and quotes the same files."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r06"
P = lambda *a: os.path.join(ROOT, *a)
NAN = float("nan")


def load(name, first_line=False):
    path = P("profiles", f"{TAG}_{name}")
    if not os.path.exists(path) or os.path.getsize(path) == 0:
        return None
    with open(path) as f:
        return json.loads(f.readline()) if first_line else json.load(f)


def put(text, new):
    a = text.index(f"<!-- {TAG}-numbers:begin")
    a = text.index("\n", a) + 1
    b = text.index(f"<!-- {TAG}-numbers:end -->")
    return text[:a] + new + text[b:]


d = load("bench.json", first_line=True)
u = load("bench_under_rocprof.json", first_line=True) or {}
pmc, mix = load("pmc_knn.json") or {}, load("knn_isa_mix.json") or {}
roll, node, pipe, longr = load("rolling_bench.json"), load("cpp_node_bench.json"), load("cpp_pipeline_bench.json"), load("long_run.json")
fe, mr, ic = load("frontend_bench.json"), load("mapreg_bench.json"), load("icp_bench.json")
ft, ftl, ft3, ft5 = load("frame_traffic.json"), load("frame_traffic_lists.json"), load("frame_traffic_c3.json"), load("frame_traffic_c5.json")
conc = load("seq_concurrency_S4.json")
R, IR, O, RP, H, pp, cb, k = (d["roofline"], d.get("issue_roofline") or {}, d["one_frame_at_a_time"], d["replay_of_preframed_maps"], d["scan_h2d_and_output"],
                              d["pose_parity_vs_cpu"], d["cpu_baseline"], d["kernel_ms_per_step"])
LZ, RU, SS, SQ, BK = d.get("lazy_target") or {}, d.get("reuse_of_an_unchanged_map") or {}, d.get("steady_state") or {}, d.get("sequences_per_gpu") or {}, d.get("roofline_by_kernel") or {}
c = {x["config"][:2]: x for x in d.get("configs", []) if "config" in x}
g = lambda o, *ks, default=NAN: (lambda v: default if v is None else v)(__import__("functools").reduce(lambda a, kk: (a or {}).get(kk) if isinstance(a, dict) else None, ks, o))
SE, LI = pmc.get("seeded") or {}, pmc.get("lists") or {}
valu = pmc.get("valu_wave_instructions_per_query", NAN)
traffic_mb = (pmc.get("hbm_bytes_per_launch") or NAN) / 1e6


def cfg_line(key, label):
    x = c.get(key)
    if not x:
        return ""
    lz = f"{x['lazy_target_scans_per_s']:.0f}" if x.get("lazy_target_scans_per_s") else "—"
    ru = f"{x['reuse_unchanged_map_scans_per_s']:.0f}" if x.get("reuse_unchanged_map_scans_per_s") else "—"
    return (f"| {label} | {x['scans_per_s']:.0f} / {x['one_frame_at_a_time_scans_per_s']:.0f} | {x['ms_per_scan']:.3g} / "
            f"{1e3 / x['one_frame_at_a_time_scans_per_s']:.3g} | {lz} | {ru} | {100 * x['hbm_frac_whole_frame']:.2f} % | {x.get('cpu_oracle_scans_per_s', NAN):.2f} | "
            f"{x.get('max_dt_m', NAN):.1e} / {x.get('max_dtheta_rad', NAN):.1e} |\n")


table = ("| config | scans/s, nothing kept between frames: two contexts / one frame at a time | ms/scan | lazy target | default reuse on the unchanged synthetic map | bytes ÷ time ÷ 8 TB/s | CPU oracle scans/s | max Δt (m) / Δθ (rad) vs oracle † |\n"
         "|---|---|---|---|---|---|---|---|\n"
         + cfg_line("c1", "c1 30 k vs 100 k, fixed map")
         + f"| **c-main 30 k vs 1 M, dependent (`value`)** | **{d['value']:.0f}** / {O['scans_per_s']:.0f} | {d['ms_per_step']} / {O['ms_per_step']} | {g(LZ, 'two_contexts', 'scans_per_s'):.0f} | "
           f"{g(RU, 'unchanged_map', 'scans_per_s'):.0f} | {100 * d['hbm_frac_whole_frame']:.2f} % | {cb['value']:.2f} at {cb['cores']} threads | {pp['max_dt_m']:.1e} / {pp['max_dtheta_rad']:.1e} ({pp['frames']} frames) |\n"
         + cfg_line("c3", "c3 HDL-64 130 k vs 5 M, dependent") + cfg_line("c5", "c5 250 k vs 20 M, dependent, IMU-like prior"))
if roll:
    A = roll["A"]
    table += (f"| c-main, map resident (`{TAG}_rolling_bench.json`; world-aligned lattice: own parity definition, DESIGN §6) | {A['resident_two_contexts_scans_per_s']:.0f} on two contexts sharing the map, "
              f"{A['resident_scans_per_s']:.0f} on one; {A['keyframe_every_3_frames_two_contexts_scans_per_s']:.0f} / {A['keyframe_every_3_frames_scans_per_s']:.0f} with a keyframe every 3rd frame | "
              f"{1e3 / A['resident_two_contexts_scans_per_s']:.3f}, {A['ms_per_frame']['resident']}; {1e3 / A['keyframe_every_3_frames_two_contexts_scans_per_s']:.3f} / {A['ms_per_frame']['keyframes']} | | | | | "
              f"{A['max_translation_diff_resident_vs_rebuild_m']:.1e} vs rebuild |\n")
table += "\n† the oracle is this repository's C restatement of the cited reference lines; the reference cannot be built here (\"parity unpinned\", DESIGN §3).\n"


def traffic_line(name, f, alg):
    if not f or not alg:
        return ""
    top = "; ".join(f"`{r['kernel'][:26]}` {r['MB_per_frame']:.0f}" for r in f["per_kernel"][:4])
    return f"{name} {f['bytes_per_frame_measured'] / 1e6:.0f} MB measured / {alg / 1e6:.0f} MB algorithmic = **{f['bytes_per_frame_measured'] / alg:.2f} ×** (largest, MB per frame: {top})"


def grid_share(f):
    """share of a frame's measured traffic that the grid build moves (the counting sort: SURVEY 8d's B has no term for it)"""
    if not f:
        return NAN
    tot = sum(r["MB_per_frame"] for r in f["per_kernel"])
    return sum(r["MB_per_frame"] for r in f["per_kernel"] if r["kernel"].startswith(("k_count", "k_rank_gather", "k_place", "k_cells_", "k_bbox"))) / tot


tl = [traffic_line("c-main", ft, d["algorithmic_bytes_per_scan"]), traffic_line("c3", ft3, g(c, "c3", "algorithmic_bytes_per_scan", default=None)),
      traffic_line("c5", ft5, g(c, "c5", "algorithmic_bytes_per_scan", default=None))]
traffic_txt = ("Frame-level HBM traffic, measured (`profiles/" + TAG + "_frame_traffic*.json`: `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes over a dependent sequence with "
               "nothing kept between frames, every kernel of a frame; FETCH_SIZE × 2 as the guide prescribes for gfx950): " + "; ".join(x for x in tl if x) + "."
               + f"  The grid build -- the counting sort that stands where the reference builds kd-trees, for which SURVEY §8d's B has no term -- moves {100 * grid_share(ft):.0f} % / {100 * grid_share(ft3):.0f} % / {100 * grid_share(ft5):.0f} % of that (c-main / c3 / c5); the kernels B does cover run at 1.3–2.0 × their algorithmic bytes (`roofline_by_kernel`)."
               + (f"  With the default reuse mode on the unchanged map the c-main frame moves {ftl['bytes_per_frame_measured'] / 1e6:.0f} MB (its lists: `{TAG}_frame_traffic_lists.json`).\n" if ftl else "\n")) if any(tl) else ""
bk_rows = ""
for r in BK.get("kernels", []):
    bk_rows += (f"| `{r['kernel'][:40]}` | {r['us_per_frame']:.1f} | {r['launches_per_frame']:.1f} | {(r.get('algorithmic_bytes_per_frame') or 0) / 1e6:.2f} | "
                f"{100 * r.get('frac_of_hbm_peak', NAN):.2f} % | {r.get('measured_MB_per_frame', NAN):.1f} | {r.get('measured_over_algorithmic', NAN)} |\n")
bk_txt = ("`roofline_by_kernel` — the five kernels with the most GPU time per frame of the timed workload (`profiles/" + TAG + "_frame_kernel_times.json`, two contexts, "
          f"{BK.get('kernel_us_per_frame_total', NAN)} µs of kernel time per frame in all), each against ITS algorithmic bytes (SURVEY §8d's terms; the kernel's own minimum where 8d has none):\n\n"
          "| kernel | µs / frame | launches / frame | algorithmic MB / frame | ÷ time ÷ 8 TB/s | measured MB / frame | measured ÷ algorithmic |\n|---|---|---|---|---|---|---|\n" + bk_rows + "\n") if bk_rows else ""
sq_txt = ""
if SQ.get("runs"):
    sq_txt = ("`sequences_per_gpu` — S independent sequences on ONE GPU, a C++ host thread each (`rgc-slam_amd/cpp/sequences_per_gpu.cpp`; every sequence's poses those of its run alone): "
              + ", ".join(f"S = {x['S']}: {x['aggregate_scans_per_s']:.0f} scans/s ({x.get('over_one_sequence', NAN):.2f} ×)" for x in SQ["runs"]) + ".")
    if conc:
        kk = {x["kernel"][:14]: x for x in conc.get("kernels", [])}
        sq_txt += (f"  The kernel trace of S = 4 (`profiles/{TAG}_seq_concurrency_S4.json`): three or more kernels in flight {100 * g(conc, 'time_with_n_kernels_in_flight', '3+'):.0f} % of the time, none "
                   f"{100 * g(conc, 'time_with_n_kernels_in_flight', '0'):.1f} %; beside the others `k_lm_step` takes {g(kk.get('k_lm_step'), 'stretch')} × as long as alone, "
                   f"`k_voxel_build_coop` {g(kk.get('k_voxel_build'), 'stretch')} ×, `k_count<true>` {g(kk.get('k_count<true>'), 'stretch')} ×, the map's kNN launch {g(kk.get('k_knn_sp<20, t'), 'stretch')} ×: "
                   "the chip is not idle while one sequence solves -- its CUs' wave slots are held by the other context's map preparation, and a second sequence's launches queue for them.")
    sq_txt += "\n"
ru_txt = ""
if RU:
    ru_txt = (f"The library's default reuse mode on THIS sequence (`reuse_of_an_unchanged_map`, each timed like `value`): unchanged map {g(RU, 'unchanged_map', 'scans_per_s'):.0f} scans/s "
              f"({g(RU, 'unchanged_map', 'queries_searched_per_launch'):.0f} of the 1 M queries searched; the launch alone {1e3 * g(RU, 'map_knn_launch_alone_ms', 'unchanged_map_lists'):.0f} µs), seeds only "
              f"{g(RU, 'seeds_only_unchanged_map', 'scans_per_s'):.0f}; one point edited every frame {g(RU, 'one_point_edited_every_frame', 'scans_per_s'):.0f} (nothing kept: "
              f"{g(RU, 'one_point_edited_every_frame', 'with_nothing_kept', 'scans_per_s'):.0f}); a keyframe (1 % of the points) every third frame {g(RU, 'keyframe_every_3rd_frame', 'scans_per_s'):.0f} "
              f"(nothing kept: {g(RU, 'keyframe_every_3rd_frame', 'with_nothing_kept', 'scans_per_s'):.0f}); the same poses as with nothing kept, bit for bit, in every case.\n")
lazy_txt = ""
if LZ:
    lazy_txt = (f"Lazy target (`lazy_target`; `value` stays the full rebuild): covariances and voxels only within {LZ.get('margin_cells')} voxels of where the scan falls at the guess -- "
                f"**{g(LZ, 'two_contexts', 'scans_per_s'):.0f} scans/s** on two contexts, {g(LZ, 'one_frame_at_a_time', 'scans_per_s'):.0f} one frame at a time, the full rebuild's poses bit for bit, "
                f"{LZ.get('solves_repeated_on_the_completed_map')} solve(s) repeated on the completed map.\n")

numbers = f'''Round-6 numbers (MI355X, `profiles/{TAG}_*`, all from one `scripts/refresh_profiles.sh` run).  **{d["value"]:.0f} registered scans/s ({d["ms_per_step"]} ms/step)** for the dependent
c-main sequence on two contexts with NOTHING kept from one frame's target to the next (`rgc_set_knn_reuse(RGC_REUSE_NONE)`: all {g(R, "queries_searched_per_launch"):.0f} map queries
searched every frame; host frame loop: {g(d, "config", "host_frame_loop", default="?")}), {O["scans_per_s"]:.0f} ({O["ms_per_step"]} ms) one frame at a time, {H["scans_per_s"]:.0f} with the scan's H2D and the
output cloud inside the step, {RP["scans_per_s"]:.0f} for the replay of pre-framed maps; repeated back to back (`steady_state`) {g(SS, "two_contexts", "scans_per_s"):.0f} / {g(SS, "one_frame_at_a_time", "scans_per_s"):.0f}
(the Python frame loop: {g(SS, "two_contexts_python_frame_loop", "scans_per_s"):.0f}).  Pose parity vs the CPU oracle over {pp["frames"]} timed frames ≤ {pp["max_dt_m"]:.1e} m / {pp["max_dtheta_rad"]:.1e} rad;
the CPU port runs {cb["value"]:.2f} scans/s at the reference's {cb["cores"]} OpenMP threads.  Round 5's headline (3475 in the driver's run) timed the default reuse mode on this
sequence's never-changing map; that figure is now the key `reuse_of_an_unchanged_map` ({g(RU, "unchanged_map", "scans_per_s"):.0f} in this run).
Per step, one frame at a time (HIP events around every stage, separate pass): grid build {k["grid_build"]:.3f} ms (both clouds), kNN map {k["knn_cov_target"]:.3f}, voxel map + the map's deferred
queries {k["voxel_build"]:.3f}, kNN scan {k["knn_cov_source"]:.3f} + {k["knn_coop_source"]:.3f} (second stream, overlapped), LM {k["linearize"]:.3f} ({d["mean_outer_iterations"]} outer iterations), fitness {k["fitness"]:.3f}.

The dominant kernel, `k_knn_sp<20, true, true, false>` (the full search): {R["avg_launch_ms"] * 1e3:.0f} µs per launch in the timed region = **{100 * R["frac"]:.2f} % of 8 TB/s** on its 36 algorithmic MB
({g(u, "roofline", "avg_launch_ms") * 1e3:.0f} µs under rocprofv3, `profiles/{TAG}_kernel_stats.csv`), {g(R, "launch_alone_ms") * 1e3:.0f} µs alone; measured HBM traffic {traffic_mb:.1f} MB per launch = {traffic_mb / 36:.2f} × algorithmic;
**{valu:.1f} VALU wave-instructions per query** ({g(SE, "valu_wave_instructions_per_query"):.1f} seeded, {g(LI, "valu_wave_instructions_per_query"):.1f} with lists; 173 in round 1), {100 * g(mix, "half_rate_fraction"):.0f} % of them in the half-rate class:
{100 * g(IR, "frac_of_mix_weighted_peak"):.0f} % of the issue rate that mix allows ({g(mix, "peak_mix_weighted"):.0f} G wave-instr/s, `profiles/{TAG}_knn_isa_mix.json`).

{bk_txt}{sq_txt}{ru_txt}{lazy_txt}{traffic_txt}
All configurations of BASELINE.json (`profiles/{TAG}_bench.json` → `configs`; target rebuilt in full every frame with nothing kept, inputs resident, first frame checked against the CPU oracle):

{table}
'''
extra = ""
if longr:
    extra += (f"Steady state (`{TAG}_long_run.json`, {longr['frames']} consecutive frames of the replay, one at a time): median {longr['ms_median']:.3f} ms, p99 {longr['ms_p99']:.3f}, "
              f"maximum {longr['ms_max']:.2f}, {longr['frames_over_1ms']} frame(s) over 1 ms, working set {longr['working_set_MiB']:.0f} MiB with "
              f"{longr['steady_state_growth_MiB_frames_200_to_end']} MiB growth between frame 200 and the end.\n")
ld = load("long_run_dependent.json")
if ld:
    a, b = ld["one_context"], ld["two_contexts"]
    extra += (f"The dependent sequence itself, default reuse mode (`{TAG}_long_run_dependent.json`: the same {ld['frames_per_repetition']} frames {ld['repetitions']} times over): "
              f"{a['scans_per_s_median']:.0f} scans/s on one context, {b['scans_per_s_median']:.0f} on two, every repetition and both modes bit-identical poses, "
              f"{a['growth_MiB_after_the_second_repetition']} / {b['growth_MiB_after_the_second_repetition']} MiB of growth after the second repetition.\n")
if node:
    extra += (f"The C++ node (`{TAG}_cpp_node_bench.json`, 24 sweeps × 28.8 k points, message bytes in → pose out; mean / median of the timed sweeps): "
              f"reference semantics {node['cpp_reference_semantics_ms_per_frame']:.2f} / {node.get('cpp_reference_semantics_median_ms', NAN):.2f} ms host-staged and "
              f"**{node['cpp_reference_semantics_device_chain_ms_per_frame']:.2f} / {node.get('cpp_reference_semantics_device_chain_median_ms', NAN):.2f} ms on the device**; "
              f"resident map {node['cpp_resident_map_ms_per_frame']:.2f} host-staged, {node['cpp_resident_map_device_chain_ms_per_frame']:.2f} with `device_chain`; "
              f"`ReplayPipeline` {node['cpp_replay_pipeline_ms_per_frame']:.2f} ms per sweep.")
    if pipe:
        extra += f" `rgc::PipelinedVGICP` on the replay workload: {pipe['pipelined_scans_per_s']:.0f} scans/s pipelined, {pipe['one_at_a_time_scans_per_s']:.0f} one at a time."
    extra += "\n"
side = []
if fe:
    side.append(f"front-end {fe['gpu_ms']:.2f} ms per VLP-16 sweep (CPU oracle, one thread: {fe['cpu_oracle_ms_1_thread']:.1f} ms)")
if mr:
    side.append(f"f1 {mr['gpu_ms_per_frame']:.2f} ms per mapping frame (CPU oracle at 14 threads: {mr['cpu_oracle_ms_14_threads']:.1f} ms)")
if ic:
    side.append(f"f4 {ic['gpu_ms']:.1f} ms per loop-closure ICP (CPU oracle: {ic['cpu_oracle_ms_14_threads']:.1f} ms)")
if side:
    extra += "Side benches of the same run: " + "; ".join(side) + ".\n"

gr = load("general_route.json")
if gr and gr.get("rows"):
    tuned = next((r for r in gr["rows"] if r["route"] == "tuned"), None)
    gen = [r for r in gr["rows"] if r["route"] == "general"]
    if tuned and gen:
        numbers += (f"The general covariance route (§5.3; `profiles/{TAG}_general_route.json`: 30 k-point scan against a 100 k-point map, host clouds in, pose out): "
                    f"{min(r['ms_per_registration'] for r in gen):.1f}–{max(r['ms_per_registration'] for r in gen):.1f} ms per registration over the "
                    f"{len(gen)} other method × mode combinations, against {tuned['ms_per_registration']:.2f} ms for PLANE / ADDITIVE on the tuned kernels; every "
                    f"combination recovers the synthetic motion to {max(r['max_abs_translation_error_vs_truth_m'] for r in gr['rows']):.3f} m.\n\n")
s = open(P("DESIGN.md")).read()
open(P("DESIGN.md"), "w").write(put(s, numbers + "\n"))

B = d["algorithmic_bytes_per_scan"] / 1e6
base = f'''Round 6, one MI355X, `python bench.py` (`profiles/{TAG}_bench.json`; everything under `profiles/{TAG}_*` is from the same run of `scripts/refresh_profiles.sh`; this block is
generated from those files by `scripts/sync_docs.py`).  HIP = this repository's gfx950 path; the target is rebuilt in full every frame and **nothing is kept from one frame's
target to the next** (`rgc_set_knn_reuse(RGC_REUSE_NONE)`): every frame pays the exact 20-NN of every map point, as the reference does and as a caller does whose map's point
set changes every frame (the reference re-filters its sub-map in the new body frame every frame, `RGC_odometer.cpp:985-991`).  c-main, c3 and c5 run as **dependent sequences** —
frame i's target is the map re-expressed on the device in the body frame of the pose frame i − 1 produced (`:1248-1256`), so only the scan's preparation can overlap the previous
solve; c1, the reference's CPU-runnable case, registers against a fixed map.  Two figures per configuration: two contexts taking turns (`value` of the bench line) / one frame
at a time through the blocking `align()`; both give bit-identical poses (checked in every run).  CPU = the C/OpenMP restatement (`oracle/`, the parity checker) on the GPU box's
host at the reference's {cb["cores"]} OpenMP threads.  The reference itself cannot be built (section 2), so there is no reference row.  Two opt-in columns, same poses bit for bit:
the lazy target (`rgc_set_target_lazy(2)`: covariances and voxels only where the solve can look) and the library's default reuse mode (seeds + neighbour lists) on the synthetic
sequences' never-changing world-frame map — a figure about that map, which is why it is not the headline (round 5 had it there: 3475 in the driver's run).

{table}
| c2 sequence stand-in (front-end + frame body + ground factor, 3 keyframes), C++ node | {f"reference semantics on the device {1e3 / node['cpp_reference_semantics_device_chain_ms_per_frame']:.0f} sweeps/s, resident map + device chain {1e3 / node['cpp_resident_map_device_chain_ms_per_frame']:.0f}, replay pipeline {1e3 / node['cpp_replay_pipeline_ms_per_frame']:.0f} (`profiles/{TAG}_cpp_node_bench.json`, 24 sweeps)" if node else "see profiles/"}; at its stated length, 200 sweeps with keyframe turnover and a trip of the ground gate: `tests/test_gpu_sequence.py::test_c2_standin_200_sweeps` | | | | | | ≤ 1e-4 vs the oracle frame body, every sweep |
| c4 8 × (30 k vs 1 M) | measured by the driver (`bench.py --gpus 8`, one sequence and two contexts per rank, no collective); on one GPU: `sequences_per_gpu` below | | | | | | |

Algorithmic bytes of a c-main scan (the formula above): B = {B:.1f} MB ⇒ {d["hbm_gbps_algorithmic"]:.0f} GB/s = {100 * d["hbm_frac_whole_frame"]:.1f} % of 8 TB/s for the whole frame.
{traffic_txt}
The dominant kernel (the map's exact 20-NN + covariance launch, 36 B per point, every query searched): {R["avg_launch_ms"] * 1e3:.0f} µs per launch in the timed region = {100 * R["frac"]:.2f} % of 8 TB/s;
{g(R, "launch_alone_ms") * 1e3:.0f} µs alone (round 4, the same search of an axis-aligned map copy: 149 µs; round 1: 353 µs); measured HBM traffic {traffic_mb:.1f} MB per launch = {traffic_mb / 36:.2f} × algorithmic
(`profiles/{TAG}_pmc_knn.json`).  The north star's "≥ 50 % of HBM roofline" is the yardstick of a streaming kernel.  This one is an exact 20-NN: per query it looks at ≈ 85
candidates (3×3×3 cells of the 1 m grid) to find 20, which costs {valu:.0f} VALU wave-instructions per query, {100 * g(mix, "half_rate_fraction"):.0f} % of them compare / select / fp64, which gfx950
issues at HALF rate (measured: 595 G wave-instr/s against 1060 for add / mul / fma, `profiles/r02_valu_issue.jsonl`); against the peak weighted by that executed mix
({g(mix, "peak_mix_weighted"):.0f} G/s) the launch runs at {100 * g(IR, "frac_of_mix_weighted_peak"):.0f} %.  Instructions per query of the full search over six rounds: 173 → 87.9 → 76.2 → 75.9 → 80.4 → {valu:.0f}.

{bk_txt}{sq_txt}{ru_txt}'''
s = open(P("BASELINE.md")).read()
open(P("BASELINE.md"), "w").write(put(s, base + "\n"))

readme = f'''* Measured on MI355X (`profiles/{TAG}_*`, one run): **{d["value"]:.0f} registered scans/s** on the headline workload — a DEPENDENT sequence: every 30 k-point scan is registered
  to the 1 M-point map re-expressed in the previous pose's body frame on the device and rebuilt in full, **nothing kept from one frame's target to the next** (every map
  query searched every frame, like a caller whose map's point set changes every frame: the reference's does) — on two contexts, {O["scans_per_s"]:.0f} one frame at a time;
  {g(LZ, "two_contexts", "scans_per_s"):.0f} with the lazy target (covariances and voxels only where the solve can look, the same poses bit for bit); {g(RU, "unchanged_map", "scans_per_s"):.0f} with the library's default
  reuse mode on this sequence's never-changing map (round 5's headline; a figure about the map, see DESIGN.md §5.1).  Pose parity against the CPU oracle ≤ {pp["max_dt_m"]:.1e} m /
  {pp["max_dtheta_rad"]:.1e} rad over {pp["frames"]} timed frames; the CPU port runs {cb["value"]:.2f} scans/s at the reference's {cb["cores"]} OpenMP threads.  c1 {g(c, "c1", "scans_per_s"):.0f}, c3 {g(c, "c3", "scans_per_s"):.0f}, c5 {g(c, "c5", "scans_per_s"):.0f} scans/s{f"; {roll['A']['resident_two_contexts_scans_per_s']:.0f} scans/s against a map resident on the device" if roll else ""}.
  The dominant kernel (exact 20-NN + covariance of the 1 M-point map, every query searched) takes {R["avg_launch_ms"] * 1e3:.0f} µs in the timed region: {100 * R["frac"]:.1f} % of the HBM roofline on its 36 algorithmic
  MB, {100 * g(IR, "frac_of_mix_weighted_peak"):.0f} % of the VALU issue rate its instruction mix allows.  S sequences on one GPU from C++ host threads: {", ".join(f"{x['S']}: {x['aggregate_scans_per_s']:.0f}" for x in SQ.get("runs", []))} scans/s.
'''
s = open(P("README.md")).read()
open(P("README.md"), "w").write(put(s, readme))

s = open(P("profiles", "README.md")).read()
prof = f'''(figures of this run: headline {d["value"]:.0f} scans/s on two contexts with nothing kept between frames, {O["scans_per_s"]:.0f} one frame at a time, replay {RP["scans_per_s"]:.0f}; the map's full-search launch
{g(R, "launch_alone_ms") * 1e3:.0f} µs alone, {R["avg_launch_ms"] * 1e3:.0f} µs in the timed region; {valu:.1f} VALU wave-instructions per map query ({g(SE, "valu_wave_instructions_per_query"):.1f} seeded,
{g(LI, "valu_wave_instructions_per_query"):.1f} with lists); {traffic_mb:.1f} MB per 1 M-query launch; default reuse mode on the unchanged map {g(RU, "unchanged_map", "scans_per_s"):.0f} scans/s)

''' + extra
open(P("profiles", "README.md"), "w").write(put(s, prof))
print("synced:", d["value"], O["scans_per_s"], RP["scans_per_s"], [x.get("scans_per_s") for x in d.get("configs", [])])
