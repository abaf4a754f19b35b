"""Rewrite the number-bearing passages of DESIGN.md (§8), BASELINE.md (§4), README.md and profiles/README.md from the files under
profiles/r02_* (one refresh = one run of scripts/refresh_profiles.sh), so that the prose cannot drift from the measurements."""
import ast, csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, *a)
J = lambda name: json.loads(open(P("profiles", name)).readline()) if name.endswith("bench.json") and "rolling" not in name and "cpp" not in name and "frontend" not in name and "mapreg" not in name and "icp" not in name and "pre_" not in name else json.load(open(P("profiles", name)))
d = json.loads(open(P("profiles", "r02_bench.json")).readline())
u = json.loads(open(P("profiles", "r02_bench_under_rocprof.json")).readline())
roll = json.load(open(P("profiles", "r02_rolling_bench.json")))["A"]
node = json.load(open(P("profiles", "r02_cpp_node_bench.json")))
pipe = json.load(open(P("profiles", "r02_cpp_pipeline_bench.json")))
longr = json.load(open(P("profiles", "r02_long_run.json"))) if open(P("profiles", "r02_long_run.json")).read().strip().startswith("{\"") else ast.literal_eval(open(P("profiles", "r02_long_run.json")).read())
pmc = json.load(open(P("profiles", "r02_pmc_knn.json")))
lp = json.load(open(P("profiles", "r02_long_run_pipelined.json")))
knn_prof_us = None
for r in csv.DictReader(open(P("profiles", "r02_kernel_stats.csv"))):
    if "k_knn_sp<20, true>" in r["Name"]:
        knn_prof_us = float(r["AverageNs"]) / 1e3
c = {x["config"][:2]: x for x in d["configs"]}
R, O, H, pp, cb = d["roofline"], d["one_frame_at_a_time"], d["scan_h2d_and_output"], d["pose_parity_vs_cpu"], d["cpu_baseline"]
k = d["kernel_ms_per_step"]
valu = pmc["valu_wave_instructions_per_query"]
traffic_mb = pmc["hbm_bytes_per_launch"] / 1e6
alone_us = R["launch_alone_ms"] * 1e3
issue_alone = valu * 1e6 / (R["launch_alone_ms"] * 1e-3) / 1e9


def sub_block(text, start, end, new):
    a = text.index(start)
    b = text.index(end, a)
    return text[:a] + new + text[b:]


def cfg_row(key, label, r1):
    x = c[key]
    return (f"| {label} | {x['scans_per_s']:.0f} / {x['one_frame_at_a_time_scans_per_s']:.0f} | {x['ms_per_scan']:.3g} / "
            f"{1e3 / x['one_frame_at_a_time_scans_per_s']:.3g} | {r1} | {x['cpu_oracle_scans_per_s']:.2f} | {x['max_dt_m']:.1e} / {x['max_dtheta_rad']:.1e} |\n")


# ---------------- DESIGN.md §8
s = open(P("DESIGN.md")).read()
new = f'''Round-2 numbers (MI355X, `profiles/r02_*`, all from one `scripts/refresh_profiles.sh` run): **{d["value"]:.0f} scans/s ({d["ms_per_step"]} ms/step) pipelined**,
{O["scans_per_s"]:.0f} scans/s ({O["ms_per_step"]} ms) one frame at a time (round 1: 1497), {H["scans_per_s"]:.0f} scans/s with the scan's H2D and the output cloud inside the step; pose
parity vs CPU over {pp["frames"]} timed frames ≤ {pp["max_dt_m"]:.1e} m / {pp["max_dtheta_rad"]:.1e} rad; CPU port {cb["value"]:.2f} scans/s at the reference's {cb["cores"]} OpenMP threads ({cb.get("value_all_cores", 0):.2f} on all host cores: more threads make it slower)
(a reported baseline, not a target). Per step, one frame at a time (HIP events, separate pass with every stage bracketed): grid build
{k["grid_build"]:.3f} ms (both clouds), kNN map {k["knn_cov_target"]:.3f} + {k["knn_coop_target"]:.3f} ms, kNN scan {k["knn_cov_source"]:.3f} + {k["knn_coop_source"]:.3f} ms (second stream, overlapped), voxel map {k["voxel_build"]:.3f}, LM {k["linearize"]:.3f}
({d["mean_outer_iterations"]} outer iterations), fitness {k["fitness"]:.3f}. `profiles/r02_kernel_stats.csv` (rocprofv3 `--kernel-trace --stats` of the same
command) agrees: `k_knn_sp<20, true>` {knn_prof_us:.0f} µs average under the profiler against {u["roofline"]["avg_launch_ms"] * 1e3:.0f} µs from the events in that run ({R["avg_launch_ms"] * 1e3:.0f} µs unprofiled:
under the profiler the host is slower and the frames overlap less); alone the launch takes {alone_us:.0f} µs. The native C++ driver
(`rgc::PipelinedVGICP`, `profiles/r02_cpp_pipeline_bench.json`) measures the same: {pipe["pipelined_scans_per_s"]:.0f} pipelined / {pipe["one_at_a_time_scans_per_s"]:.0f} one at a time — the host
language is not what bounds the loop. (Box to box these figures move by ±3 %.)

Steady state (`profiles/r02_long_run.json`, {longr["frames"]} consecutive frames one at a time): median {longr["ms_median"]:.3f} ms, p99 {longr["ms_p99"]:.3f}, maximum {longr["ms_max"]:.2f}, {longr["frames_over_1ms"]} frame(s)
over 1 ms, working set {longr["working_set_MiB"]:.0f} MiB with {longr["steady_state_growth_MiB_frames_200_to_end"]} MiB growth between frame 200 and the end, every repetition of an input gives the bit-identical
pose (fixed-order folds, deterministic cell order). Through the two-context pipeline (`profiles/r02_long_run_pipelined.json`, {lp["frames"]} frames):
{lp["scans_per_s_overall"]:.0f} scans/s overall, {lp["scans_per_s_per_500_frames_min_max"][0]:.0f}–{lp["scans_per_s_per_500_frames_min_max"][1]:.0f} per block of 500 frames, {lp["working_set_MiB"]:.0f} MiB for the two contexts, {"no" if lp["growth_MiB_frames_400_to_end"] == 0 else str(lp["growth_MiB_frames_400_to_end"]) + " MiB of"} growth, identical poses on every repetition.

All configurations of BASELINE.json (`profiles/r02_bench.json` → `configs`; target rebuilt every frame, inputs resident, first frame
checked against the CPU oracle; pipelined / one at a time):

| config | scans/s | ms/scan | round 1 | CPU oracle scans/s | max Δt (m) / Δθ (rad) vs oracle |
|---|---|---|---|---|---|
''' + cfg_row("c1", "c1 30 k vs 100 k (15 outer iterations from the identity)", 1386) + \
    f'''| c-main 30 k vs 1 M | {d["value"]:.0f} / {O["scans_per_s"]:.0f} | {d["ms_per_step"]} / {O["ms_per_step"]} | 1497 | {cb["value"]:.2f} at {cb["cores"]} threads ({cb.get("value_all_cores", 0):.2f} on all cores) | {pp["max_dt_m"]:.1e} / {pp["max_dtheta_rad"]:.1e} |
''' + cfg_row("c3", "c3 HDL-64 130 k vs 5 M", 351) + cfg_row("c5", "c5 250 k vs 20 M, IMU-preintegrated prior", 113) + \
    f'''| c-main, map resident on the device (`profiles/r02_rolling_bench.json`) | {roll["resident_two_contexts_scans_per_s"]:.0f} resident on two contexts, {roll["resident_scans_per_s"]:.0f} on one; {roll["keyframe_every_3_frames_two_contexts_scans_per_s"]:.0f} / {roll["keyframe_every_3_frames_scans_per_s"]:.0f} with a keyframe every 3rd frame | {1e3 / roll["resident_two_contexts_scans_per_s"]:.3f}, {roll["ms_per_frame"]["resident"]}; {1e3 / roll["keyframe_every_3_frames_two_contexts_scans_per_s"]:.3f} / {roll["ms_per_frame"]["keyframes"]} | 2061, 1402 | | {roll["max_translation_diff_resident_vs_rebuild_m"]:.1e} vs rebuild |

'''
s = sub_block(s, "Round-2 numbers (MI355X, `profiles/r02_*`", "History (scans/s on c-main", new)
# the same figures where §5 and §6e quote them
s = re.sub(r"\d+ → \d+ scans/s in `bench.py` \(\d+ → [\d–]+ in the bare drivers `scripts/exp_pipeline.py` and `bench_cpp_pipeline.py`\); c1 \(30 k vs\n100 k, 15 outer iterations\): \d+ → \d+; c3: \d+ → \d+; c5: \d+ → \d+\.",
           f"{O['scans_per_s']:.0f} → {d['value']:.0f} scans/s in `bench.py` ({pipe['one_at_a_time_scans_per_s']:.0f} → {pipe['pipelined_scans_per_s']:.0f} in the bare drivers `scripts/exp_pipeline.py` and `bench_cpp_pipeline.py`); c1 (30 k vs\n"
           f"100 k, 15 outer iterations): {c['c1']['one_frame_at_a_time_scans_per_s']:.0f} → {c['c1']['scans_per_s']:.0f}; c3: {c['c3']['one_frame_at_a_time_scans_per_s']:.0f} → {c['c3']['scans_per_s']:.0f}; c5: {c['c5']['one_frame_at_a_time_scans_per_s']:.0f} → {c['c5']['scans_per_s']:.0f}.", s)
s = re.sub(r"source: \*\*\d+ scans/s against \d+ on one context\*\*", f"source: **{roll['resident_two_contexts_scans_per_s']:.0f} scans/s against {roll['resident_scans_per_s']:.0f} on one context**", s)
s = re.sub(r"three frames\): \d+ against \d+ scans/s\.", f"three frames): {roll['keyframe_every_3_frames_two_contexts_scans_per_s']:.0f} against {roll['keyframe_every_3_frames_scans_per_s']:.0f} scans/s.", s)
# the side benches quoted in §6 (front-end, C++ node, f1, f4, f2)
fe = json.load(open(P("profiles", "r02_frontend_bench.json")))
mr = json.load(open(P("profiles", "r02_mapreg_bench.json")))
ic = json.load(open(P("profiles", "r02_icp_bench.json")))
rb = json.load(open(P("profiles", "r02_rolling_bench.json")))["B"]
sel_us = hs_us = None
for r in csv.DictReader(open(P("profiles", "r02_frontend_kernel_stats.csv"))):
    if "k_fe_select" in r["Name"]:
        sel_us = float(r["AverageNs"]) / 1e3
s = re.sub(r"Measured \(`profiles/r0\d_frontend_bench.json`\): \*\*[\d.–]+ ms per VLP-16 sweep\*\* on MI355X — two host round trips\n\(ring counts; [^)]*\) — against [\d.]+ ms for the single-threaded CPU oracle; the selection\nkernel is \d+ µs of it",
           f"Measured (`profiles/r02_frontend_bench.json`): **{fe['gpu_ms']:.2f} ms per VLP-16 sweep** on MI355X — two host round trips\n(ring counts; then ground sums, flags and the three feature clouds in ONE copy of a device tail laid out like the pinned staging area) — against {fe['cpu_oracle_ms_1_thread']:.1f} ms for the single-threaded CPU oracle; the selection\nkernel is {sel_us:.0f} µs of it", s)
s = re.sub(r"Measured \(`profiles/r0\d_cpp_node_bench.json`, 24 sweeps × 28.8 k points,\nmessage bytes in → pose out\): \*\*[\d.]+ ms per frame\*\* with the reference's local-map semantics, \*\*[\d.]+ ms\*\* with the resident map and\n\*\*[\d.]+ ms\*\* with `device_chain`",
           f"Measured (`profiles/r02_cpp_node_bench.json`, 24 sweeps × 28.8 k points,\nmessage bytes in → pose out): **{node['cpp_reference_semantics_ms_per_frame']:.2f} ms per frame** with the reference's local-map semantics, **{node['cpp_resident_map_ms_per_frame']:.2f} ms** with the resident map and\n**{node['cpp_resident_map_device_chain_ms_per_frame']:.2f} ms** with `device_chain`", s)
s = re.sub(r"— \d+–\d+ frames/s against the sensor's\n10 Hz \(Python mirrors: [\d.]+ / [\d.]+ ms\)\. Half of a frame is the front-end, whose greedy per-sector selection runs on 16 workgroups \(\d+ µs\)\.",
           f"— {1e3 / node['cpp_reference_semantics_ms_per_frame']:.0f}–{1e3 / node['cpp_resident_map_device_chain_ms_per_frame']:.0f} frames/s against the sensor's\n10 Hz (Python mirrors: {node['python_reference_semantics_ms_per_frame']:.1f} / {node['python_resident_map_ms_per_frame']:.1f} ms). Half of a frame is the front-end, whose greedy per-sector selection runs on 16 workgroups ({sel_us:.0f} µs).", s)
s = re.sub(r"\*\*[\d.]+ ms per sweep \(\d+ sweeps/s\)\*\*, poses and ground messages identical to the unpipelined node",
           f"**{node['cpp_replay_pipeline_ms_per_frame']:.2f} ms per sweep ({1e3 / node['cpp_replay_pipeline_ms_per_frame']:.0f} sweeps/s)**, poses and ground messages identical to the unpipelined node", s)
s = re.sub(r"Measured \(`profiles/r0\d_mapreg_bench.json`, 1050 \+ 2998 / 1087 \+ 2959 features, 10 k / 8 k map points, maps re-gridded every\nframe\): \*\*[\d.]+ ms per frame on MI355X\*\* against [\d.]+ ms for the CPU oracle at the reference's 14 threads",
           f"Measured (`profiles/r02_mapreg_bench.json`, 1050 + 2998 / 1087 + 2959 features, 10 k / 8 k map points, maps re-gridded every\nframe): **{mr['gpu_ms_per_frame']:.2f} ms per frame on MI355X** against {mr['cpu_oracle_ms_14_threads']:.1f} ms for the CPU oracle at the reference's 14 threads", s)
s = re.sub(r"Measured \(`profiles/r0\d_icp_bench.json`\): 20 k-point\nkey frame against a 200 k-point history sub-map, 25 iterations: \*\*[\d.]+ ms on MI355X\*\*, [\d.]+ ms for the CPU oracle at 14 threads\.",
           f"Measured (`profiles/r02_icp_bench.json`): 20 k-point\nkey frame against a 200 k-point history sub-map, 25 iterations: **{ic['gpu_ms']:.1f} ms on MI355X**, {ic['cpu_oracle_ms_14_threads']:.1f} ms for the CPU oracle at 14 threads.", s)
open(P("DESIGN.md"), "w").write(s)

# ---------------- BASELINE.md §4
s = open(P("BASELINE.md")).read()
B = d["algorithmic_bytes_per_scan"] / 1e6


def brow(key, label, r1):
    x = c[key]
    return (f"| {label} | CPU | 14 | {x['cpu_oracle_scans_per_s']:.2f} | | | |\n"
            f"| {label} | HIP | 1 GPU | **{x['scans_per_s']:.0f}** pipelined / {x['one_frame_at_a_time_scans_per_s']:.0f} one at a time (round 1: {r1}) | "
            f"{x['ms_per_scan']:.3g} / {1e3 / x['one_frame_at_a_time_scans_per_s']:.3g} | {x['max_dt_m']:.1e} | {x['max_dtheta_rad']:.1e} |\n")


new = f'''## 4. Results

Round 2, one MI355X, `python bench.py --configs c1,c3,c5` (`profiles/r02_bench.json`; everything under `profiles/r02_*` is from the same
run of `scripts/refresh_profiles.sh`; this section is generated from those files by `scripts/sync_docs.py`). HIP = this repository's
gfx950 path, target rebuilt every frame, inputs resident in HBM; two figures per configuration: **pipelined** (two contexts take turns:
frame i + 1's clouds are prepared while frame i is solved — the throughput of a replayed sequence, `value` of the bench line) and **one
frame at a time** (the blocking `align()`: a frame's latency). Both give bit-identical poses (checked in every run). CPU = the C/OpenMP
restatement (`oracle/`, the parity checker) on the GPU box's host at the reference's {cb["cores"]} OpenMP threads (c-main also on all cores, in
brackets: slower). The reference itself cannot be built (section 2), so there is no reference row.

| Config | Backend | Threads / GPUs | scans/s | ms/scan | max Δt (m) vs CPU | max Δθ (rad) |
|---|---|---|---|---|---|---|
''' + brow("c1", "c1 30 k vs 100 k", 1386) + \
    f'''| c2 sequence stand-in (24 sweeps × 28.8 k pts, front-end + frame body + ground factor, 3 keyframes) | HIP, C++ node | 1 GPU | {1e3 / node["cpp_reference_semantics_ms_per_frame"]:.0f} (reference semantics) / {1e3 / node["cpp_resident_map_device_chain_ms_per_frame"]:.0f} (resident map, device chain) / {1e3 / node["cpp_replay_pipeline_ms_per_frame"]:.0f} (replay pipeline) (`profiles/r02_cpp_node_bench.json`) | {node["cpp_reference_semantics_ms_per_frame"]:.2f} / {node["cpp_resident_map_device_chain_ms_per_frame"]:.2f} / {node["cpp_replay_pipeline_ms_per_frame"]:.2f} | ≤ 1e-4 vs the oracle frame body, with and without the IMU path (`tests/test_gpu_sequence.py`, `tests/test_gpu_cpp_node.py`) | ≤ 1e-4 |
| c-main 30 k vs 1 M | CPU | {cb["cores"]} (all cores) | {cb["value"]:.2f} ({cb.get("value_all_cores", 0):.2f}) | | | |
| c-main 30 k vs 1 M | HIP | 1 GPU | **{d["value"]:.0f}** pipelined / {O["scans_per_s"]:.0f} one at a time / {H["scans_per_s"]:.0f} pipelined with the scan uploaded from pinned memory inside the step (round 1: 1497) | {d["ms_per_step"]} / {O["ms_per_step"]} / {H["ms_per_step"]} | {pp["max_dt_m"]:.1e} ({pp["frames"]} frames) | {pp["max_dtheta_rad"]:.1e} |
| c-main, C++ host (`rgc::PipelinedVGICP`) | HIP | 1 GPU | {pipe["pipelined_scans_per_s"]:.0f} pipelined / {pipe["one_at_a_time_scans_per_s"]:.0f} one at a time (`profiles/r02_cpp_pipeline_bench.json`) | {pipe["pipelined_ms_per_frame"]:.3f} / {pipe["one_at_a_time_ms_per_frame"]:.3f} | identical to the Python mirror | |
| c-main, map resident (SURVEY §8f f2) | HIP | 1 GPU | {roll["resident_two_contexts_scans_per_s"]:.0f} resident on two contexts sharing the map / {roll["resident_scans_per_s"]:.0f} on one / {roll["rebuild_every_frame_scans_per_s"]:.0f} rebuilt per frame / {roll["keyframe_every_3_frames_two_contexts_scans_per_s"]:.0f} with a keyframe every 3rd frame (`profiles/r02_rolling_bench.json`) | {1e3 / roll["resident_two_contexts_scans_per_s"]:.3f} / {roll["ms_per_frame"]["resident"]} / {roll["ms_per_frame"]["rebuild"]} / {1e3 / roll["keyframe_every_3_frames_two_contexts_scans_per_s"]:.3f} | {roll["max_translation_diff_resident_vs_rebuild_m"]:.1e} vs rebuild | |
''' + brow("c3", "c3 HDL-64 130 k vs 5 M", 351) + \
    '''| c4 8 × (30 k vs 1 M) | HIP | 8 GPUs | measured by the driver (`bench.py --gpus 8`, one sequence and two contexts per rank, no collective) | | | |
''' + brow("c5", "c5 250 k vs 20 M + IMU-preintegrated prior", 113) + f'''
Algorithmic bytes of a c-main scan (the formula above): B = {B:.1f} MB ⇒ {d["hbm_gbps_algorithmic"]:.0f} GB/s = {100 * d["hbm_frac_whole_frame"]:.1f} % of 8 TB/s for the whole frame.
The dominant kernel (the map's bulk kNN + covariance launch, 36 B per point): {R["avg_launch_ms"] * 1e3:.0f} µs per launch in the timed region
(it shares the chip with another frame's solve and the scan's kernels) = {100 * R["frac"]:.2f} % of 8 TB/s; {alone_us:.0f} µs alone =
{100 * R["frac_launch_alone"]:.2f} % (round 1: 353 µs, 1.27 %); measured HBM traffic {traffic_mb:.1f} MB per launch = {traffic_mb / 36:.1f} × algorithmic (`profiles/r02_pmc_knn.json`;
1.2 × with the alternative cell order of DESIGN.md §5, which costs 3 % of the launch's time and is not the default).

The north star's "≥ 50 % of HBM roofline" is the yardstick of a streaming kernel. This path's dominant kernel is an exact 20-NN: per
query it looks at ≈ 190 candidates (3×3×3 cells of a 1 m grid) and keeps the 22 best, which costs {valu:.0f} VALU wave-instructions per query —
most of them compare / select / `med3`, which gfx950 issues at HALF rate (measured: 595 G wave-instr/s against 1060 for add / mul / fma,
`profiles/r02_valu_issue.jsonl`). Alone the launch sustains {valu:.1f} M / {R["launch_alone_ms"]:.3f} ms = {issue_alone:.0f} G/s, {100 * issue_alone / 595:.0f} % of that measured roof; what moves the
number is therefore fewer instructions per candidate or fewer candidates per query (round 1 → 2: 173 → {valu:.0f} per query, 353 → {alone_us:.0f} µs),
not bytes. `issue_roofline` in the bench line prices the timed-region launch against both measured rates (DESIGN.md §5).
'''
s = s[:s.index("## 4. Results")] + new
open(P("BASELINE.md"), "w").write(s)

# ---------------- README.md: the measured bullet
s = open(P("README.md")).read()
new = f'''* Measured on MI355X (`profiles/r02_*`, one run): **{d["value"]:.0f} registered scans/s** on the headline workload (30 k-point scan against a
  1 M-point map, everything rebuilt per frame) with two contexts taking turns (`registration.PipelinedVGICP`: the next frame's clouds are
  prepared while a frame is solved; identical poses; `rgc::PipelinedVGICP` in C++ measures the same), {O["scans_per_s"]:.0f} scans/s one frame at a time
  (round 1: 1497); pose parity against the CPU oracle ≤ {pp["max_dt_m"]:.1e} m / {pp["max_dtheta_rad"]:.1e} rad over {pp["frames"]} timed frames; the CPU port runs {cb["value"]:.2f} scans/s at
  the reference's {cb["cores"]} OpenMP threads ({cb.get("value_all_cores", 0):.2f} on all host cores). c1 {c["c1"]["scans_per_s"]:.0f}, c3 {c["c3"]["scans_per_s"]:.0f}, c5 {c["c5"]["scans_per_s"]:.0f} scans/s; {roll["resident_two_contexts_scans_per_s"]:.0f} scans/s against a map resident on the device.
  The dominant kernel (exact 20-NN + covariance of the 1 M-point map) takes {alone_us:.0f} µs alone, {valu:.0f} VALU wave-instructions per query, at
  {100 * issue_alone / 595:.0f} % of the measured half-rate VALU issue roof (DESIGN.md §5).
'''
s = sub_block(s, "* Measured on MI355X (`profiles/r02_*`, one run):", "* The host side in the reference's language:", new)
open(P("README.md"), "w").write(s)
# ---------------- profiles/README.md: the figures quoted in its table
s = open(P("profiles", "README.md")).read()
s = re.sub(r"pose parity of \d+ frames", f"pose parity of {pp['frames']} frames", s)
s = re.sub(r"average, \d+ µs, against \d+ µs from the library's own HIP events in that run; \d+ µs in the unprofiled run:",
           f"average, {knn_prof_us:.0f} µs, against {u['roofline']['avg_launch_ms'] * 1e3:.0f} µs from the library's own HIP events in that run; {R['avg_launch_ms'] * 1e3:.0f} µs in the unprofiled run:", s)
s = re.sub(r"→ [\d.]+ VALU wave-instructions per map query, [\d.]+ MB per 1 M-query launch", f"→ {valu:.1f} VALU wave-instructions per map query, {traffic_mb:.1f} MB per 1 M-query launch", s)
s = re.sub(r"median [\d.]+ ms, p99 [\d.]+, max [\d.]+[^,]*,", f"median {longr['ms_median']:.3f} ms, p99 {longr['ms_p99']:.3f}, max {longr['ms_max']:.2f} ({longr['frames_over_1ms']} frame(s) over 1 ms),", s, count=1)
s = re.sub(r"rebuilt every frame \d+ scans/s, resident \d+ on one context and \d+ on two sharing the map \(`rgc_share_target`\), a keyframe every 3rd frame \d+ / \d+ \(commit [\d.]+ ms\);",
           f"rebuilt every frame {roll['rebuild_every_frame_scans_per_s']:.0f} scans/s, resident {roll['resident_scans_per_s']:.0f} on one context and {roll['resident_two_contexts_scans_per_s']:.0f} on two sharing the map (`rgc_share_target`), a keyframe every 3rd frame {roll['keyframe_every_3_frames_scans_per_s']:.0f} / {roll['keyframe_every_3_frames_two_contexts_scans_per_s']:.0f} (commit {roll['commit_ms_median']:.2f} ms);", s)
s = re.sub(r"[\d.]+ / [\d.]+ / [\d.]+ ms per frame \(reference semantics / resident map / device chain\), [\d.]+ ms through `ReplayPipeline`",
           f"{node['cpp_reference_semantics_ms_per_frame']:.2f} / {node['cpp_resident_map_ms_per_frame']:.2f} / {node['cpp_resident_map_device_chain_ms_per_frame']:.2f} ms per frame (reference semantics / resident map / device chain), {node['cpp_replay_pipeline_ms_per_frame']:.2f} ms through `ReplayPipeline`", s)
s = re.sub(r"\d+ scans/s pipelined, \d+ one at a time, same poses and fitness", f"{pipe['pipelined_scans_per_s']:.0f} scans/s pipelined, {pipe['one_at_a_time_scans_per_s']:.0f} one at a time, same poses and fitness", s)
open(P("profiles", "README.md"), "w").write(s)
# ---------------- README.md: the C++ node line
s = open(P("README.md")).read()
s = re.sub(r"[\d.]+ ms per 28\.8 k-point sweep with the sweep kept on the device between the stages, [\d.]+ ms in",
           f"{node['cpp_resident_map_device_chain_ms_per_frame']:.2f} ms per 28.8 k-point sweep with the sweep kept on the device between the stages, {node['cpp_replay_pipeline_ms_per_frame']:.2f} ms in", s)
s = re.sub(r"\d+ scans/s against a\n  resident 1 M-point map with two contexts sharing it, \d+ with one\)",
           f"{roll['resident_two_contexts_scans_per_s']:.0f} scans/s against a\n  resident 1 M-point map with two contexts sharing it, {roll['resident_scans_per_s']:.0f} with one)", s)
open(P("README.md"), "w").write(s)
print("synced:", d["value"], O["scans_per_s"], [x["scans_per_s"] for x in d["configs"]])
