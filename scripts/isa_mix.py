"""Executed VALU instruction mix of the map's bulk kNN kernel: full-rate vs half-rate classes.  Two instances: k_knn_sp<20, true, true, false>
(the full search: a map the library has not seen) and, with --seeded, k_knn_sp<20, true, true, true> (round 5: the search that starts from
the previous search's k-th distances, what a re-framed persistent map runs from its second frame on; the full search inlined behind it
for the lanes the seeded one declines is weighted by how many waves took it).

The kernel's ISA comes from the product's own compile flags (hipcc --save-temps, gfx950); every v_* instruction is put into the issue
class MEASURED for it by scripts/ubench/valu_issue.hip (profiles/r02_valu_issue.jsonl: add / sub / mul / fma / fmac f32, add / sub u32,
and / or / xor, mov issue at ~1060 G wave-instructions/s chip-wide; min / max / med3 / compare / select / shifts / three-operand integer
ops / packed fp32 / every fp64 at ~595 G/s); opcodes the micro-benchmark did not time are listed and priced BOTH ways.  Basic blocks are
weighted by how often they run per wave: once, or -- inside one of the kernel's wave-level loops -- by the loop's trip count measured
with a developer build (RGC_EXTRA_FLAGS=-DRGC_LAB, rgc_lab_iters: quads scanned, chain inserts, Newton steps per wave).  The result
is checked against SQ_INSTS_VALU of the rocprofv3 --pmc pass.

    python scripts/isa_mix.py --lab gpurun_out/lab_iters.json [--pmc profiles/r03_pmc_knn.json] > profiles/r03_knn_isa_mix.json
    python scripts/isa_mix.py --collect      # on the GPU box, library built with -DRGC_LAB: writes gpurun_out/lab_iters.json
"""
import argparse, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN4rgck8k_knn_spILi20ELb1ELb1ELb0E"
KERNEL_SEEDED = "_ZN4rgck8k_knn_spILi20ELb1ELb1ELb1E"
FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32"}
HALF_PREFIX = ("v_min", "v_max", "v_med3", "v_cmp", "v_cndmask", "v_lshl", "v_lshr", "v_ashr", "v_and_or", "v_bfi", "v_bfe", "v_add3", "v_mad_",
               "v_mul_lo", "v_mul_hi", "v_pk_", "v_xad", "v_addc", "v_subb", "v_add_co", "v_sub_co", "v_or3", "v_xor3", "v_lshl_or", "v_lshl_add",
               "v_add_lshl", "v_perm", "v_alignbit")


def collect():
    """GPU box, -DRGC_LAB build: trip counts of the wave-level loops over one preparation of the c-main map"""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import registration, _lib
    lib = _lib.load()
    lib.rgc_lab_iters.argtypes = [C.c_void_p, C.c_void_p]
    world, tgt = synth.make_world_and_map(1000000, seed=synth.SEED)
    v = registration.odometer_vgicp(0)
    it = np.zeros(8, np.uint64)
    v.setInputTarget(tgt); v.synchronize()
    lib.rgc_lab_iters(v._h, it.ctypes.data)      # (first preparation: resets)
    v.setInputTarget(tgt); v.synchronize()
    lib.rgc_lab_iters(v._h, it.ctypes.data)
    out = {"n_target": int(len(tgt)), "waves": int(it[0]), "quads_in_scan_loop": int(it[1]), "chain_inserts": int(it[2]), "newton_steps": int(it[3]),
           "jacobi_fallbacks": int(it[4]), "exact_tie_breaks": int(it[5])}
    # the seeded search: the same map handed over by rgc_set_target_reframed three times (poses of the bench's trajectory), the third counted
    import bench
    a = np.zeros((len(tgt), 4), np.float32); a[:, :3] = tgt
    d_map, d_body = v.device_alloc(a.nbytes), v.device_alloc(a.nbytes)
    v.upload(d_map, a)
    poses = synth.make_trajectory(4, seed=synth.SEED)
    for f in range(3):
        q, t = bench.world_to_body(np.asarray(poses[f], np.float64))
        lib.rgc_lab_iters(v._h, it.ctypes.data)
        v.setInputTargetReframed(d_map, len(tgt), 16, q, t, d_body); v.synchronize()
    lib.rgc_lab_iters(v._h, it.ctypes.data)
    out["seeded"] = {"waves": (len(tgt) + 63) // 64, "waves_that_ran_the_full_search_too": int(it[6]), "quads_in_seeded_scan_loop": int(it[7]),
                     "quads_in_scan_loop": int(it[1]), "chain_inserts": int(it[2]), "newton_steps": int(it[3]), "jacobi_fallbacks": int(it[4]),
                     "exact_tie_breaks": int(it[5])}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "lab_iters.json"), "w"))
    print(json.dumps(out))
    v.close()


def kernel_isa(KERNEL=KERNEL, extra=()):
    d = tempfile.mkdtemp(prefix="rgc_isa_")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-DRGC_BUILD", "--save-temps"] + list(extra)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(ROOT, "rgc-slam_amd", "csrc", "rgc_kernels.hip"), "-o", os.path.join(d, "k.o")],
                          cwd=d, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "rgc_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().endswith(":") or (l.startswith(KERNEL) and ": ;" in l))
    b = next(i for i in range(a, len(lines)) if lines[i].lstrip().startswith(".size") and KERNEL in lines[i])
    return lines[a:b]


def classify(op):
    if op in FULL:
        return "full"
    if op.endswith("_f64") or "f64" in op:
        return "half"
    if op.startswith(HALF_PREFIX):
        return "half"
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--collect", action="store_true")
    ap.add_argument("--lab", default=os.path.join(ROOT, "gpurun_out", "lab_iters.json"))
    ap.add_argument("--pmc", default=None)
    ap.add_argument("--seeded", action="store_true", help="the seeded instance (k_knn_sp<20, true, true, true>), lab counts from lab['seeded']")
    args = ap.parse_args()
    if args.collect:
        return collect()
    lab = json.load(open(args.lab))
    if args.seeded:
        lab = lab["seeded"]
    W = float(lab["waves"])
    # (the seeded SEARCH by itself: the neighbour-list workgroups compiled out -- with them the search is instantiated twice, once inside
    # the loop over the todo lists, and a map that is searched in full runs the copy outside it; lab counts collected under RGC_KNN_CACHE=0)
    isa = kernel_isa(KERNEL_SEEDED, ["-DRGC_KNN_CACHE=0"]) if args.seeded else kernel_isa(KERNEL)
    # ---- basic blocks ----
    blocks, cur = [], {"label": "entry", "ops": [], "branches": [], "line": 0}
    for ln, l in enumerate(isa):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ops": [], "branches": [], "line": ln}
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if not op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            cur["ends"] = False
        if op.startswith("v_"):
            cur["ops"].append(re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", op))
        elif op.startswith(("s_cbranch", "s_branch")) and len(t) > 1:   # a branch ends its basic block: what follows is an unlabeled one
            cur["branches"].append(t[1])
            cur["ends"] = op == "s_branch"
            blocks.append(cur)
            cur = {"label": "%s+%d" % (cur["label"].split("+")[0], ln), "ops": [], "branches": [], "line": ln}
        elif op in ("s_endpgm", "s_setpc_b64"):
            cur["ends"] = True
        elif op.startswith(("global_load", "global_store", "ds_", "global_atomic")):
            cur.setdefault("mem", []).append(op)
    blocks.append(cur)
    index = {b["label"]: i for i, b in enumerate(blocks)}
    # ---- control-flow graph: branch targets + fall-through (unless the block ends in an unconditional branch or the program's end) ----
    succ = []
    for i, b in enumerate(blocks):
        s_ = [index[t] for t in b["branches"] if t in index]
        if not b.get("ends", False) and i + 1 < len(blocks):
            s_.append(i + 1)
        succ.append(sorted(set(s_)))
    # ---- loops = strongly connected components (Tarjan, iterative) ----
    n = len(blocks)
    idx, low, on, st, comp, cnt = [-1] * n, [0] * n, [False] * n, [], [-1] * n, [0]
    ncomp = 0
    for r in range(n):
        if idx[r] != -1:
            continue
        work = [(r, 0)]
        while work:
            v_, pi = work[-1]
            if pi == 0:
                idx[v_] = low[v_] = cnt[0]; cnt[0] += 1
                st.append(v_); on[v_] = True
            if pi < len(succ[v_]):
                work[-1] = (v_, pi + 1)
                w_ = succ[v_][pi]
                if idx[w_] == -1:
                    work.append((w_, 0))
                elif on[w_]:
                    low[v_] = min(low[v_], idx[w_])
            else:
                work.pop()
                if work:
                    u_ = work[-1][0]
                    low[u_] = min(low[u_], low[v_])
                if low[v_] == idx[v_]:
                    while True:
                        w_ = st.pop(); on[w_] = False; comp[w_] = ncomp
                        if w_ == v_:
                            break
                    ncomp += 1
    members = {}
    for i, c_ in enumerate(comp):
        members.setdefault(c_, []).append(i)
    cyclic = {c_: m for c_, m in members.items() if len(m) > 1 or m[0] in succ[m[0]]}

    def has_mem(m, what):
        return any(what in x for i in m for x in blocks[i].get("mem", []))
    # per-wave trip counts (developer build): the scan loop's body holds two quads; chain inserts are the self-looping blocks made of
    # v_med3_i32 (the copies of the drain loop have identical bodies: only the sum of their trips matters, it is split evenly);
    # Newton steps to the loop with v_rcp_f64; the Jacobi fallback's sweeps (a handful of waves) six per fallback
    trips = {"scan": lab["quads_in_scan_loop"] / W / 2.0, "drain": lab["chain_inserts"] / W, "newton": lab["newton_steps"] / W,
             "other": 6.0 * lab["jacobi_fallbacks"] / W}
    # The seeded instance: its own candidate loop fetches 12-byte records (global_load_dwordx3), the full search inlined behind it 16-byte
    # ones; everything from the full search's entry on (the second group of cell_coord blocks) runs for the waves that took it only.
    fallback_from, fallback_w = len(blocks), 0.0
    if args.seeded:
        trips["scan_seeded"] = lab["quads_in_seeded_scan_loop"] / W / 2.0
        fallback_w = lab["waves_that_ran_the_full_search_too"] / W
        cc = [i for i, b in enumerate(blocks) if "cell_coord" in isa[b["line"]]]
        groups = [i for j, i in enumerate(cc) if j == 0 or i - cc[j - 1] > 40]
        if len(groups) >= 2:
            fallback_from = groups[1] - 8   # (the full search's prologue: a few blocks before its first cell_coord block)
    def returns_soon(i):   # back at block i within three hops: the one-block body of a drain loop (split at its own branches)
        seen, front = set(), {i}
        for _ in range(3):
            front = {w_ for v_ in front for w_ in succ[v_]}
            if i in front:
                return True
            seen |= front
        return False
    # (a drain call's first insert may be peeled out of its self-loop and then sits in the scan loop's component: every block that
    # holds a chain insert -- 22 v_med3_i32 -- is one copy of the same body, and LAB_COUNT(2) counts the inserts of all of them)
    drain_blocks = [i for i, b in enumerate(blocks) if sum(o == "v_med3_i32" for o in b["ops"]) >= 15]
    looping_drain_blocks = [i for i in drain_blocks if returns_soon(i)]
    weight = [1.0] * n
    kinds = {}
    for c_, m in cyclic.items():
        ops = [o for i in m for o in blocks[i]["ops"]]
        if args.seeded and has_mem(m, "global_load_dwordx3") and min(m) < fallback_from:
            k = "scan_seeded"
        elif has_mem(m, "global_load_dwordx3") or has_mem(m, "global_load_dwordx4"):
            k = "scan"
        elif any(o.startswith("v_rcp_f64") for o in ops) and not any(o.startswith(("v_sqrt_f64", "v_rsq_f64")) for o in ops):
            k = "newton"   # (a rolled Newton loop; the Jacobi fallback's sweeps also divide, but take roots as well)
        elif len(m) == 1 and m[0] in looping_drain_blocks:
            k = "drain"
        else:
            k = "other"
        kinds[c_] = k
        for i in m:
            weight[i] = trips[k] if k != "drain" else 0.0
    for i in drain_blocks:
        weight[i] = trips["drain"] / len(drain_blocks)
    if args.seeded:   # the inlined full search: its once-per-wave blocks run for the waves that took it; its loops are weighted by their own counts (of those waves)
        for i in range(max(fallback_from, 0), n):
            if weight[i] == 1.0:
                weight[i] = fallback_w
    # The Newton iteration of min_eigenvector_direct is fully unrolled (twelve steps, each leaving through an exec-mask branch): step j of the
    # chain runs for the waves that still have an unsettled lane; with `newton` steps per wave on average, step j gets clamp(newton - j, 0, 1)
    chain = [i for i, b in enumerate(blocks) if sum(o.startswith("v_rcp_f64") for o in b["ops"]) == 1 and b["branches"] and comp[i] not in cyclic
             and not any(o.startswith("v_div_") for o in b["ops"])]
    if len(chain) >= 8:
        for j, i in enumerate(chain):
            weight[i] = min(1.0, max(0.0, trips["newton"] - j))
        kinds["unrolled"] = "newton"
    # (blocks outside every loop are weighted 1 per wave, see the note in the output)
    tot = {"full": 0.0, "half": 0.0, "unknown": 0.0}
    static = {"full": 0, "half": 0, "unknown": 0}
    per_op = {}
    for b, w in zip(blocks, weight):
        for o in b["ops"]:
            c = classify(o)
            tot[c] += w
            static[c] += 1
            d = per_op.setdefault(o, [c, 0, 0.0])
            d[1] += 1
            d[2] += w
    if os.environ.get("ISA_MIX_DEBUG"):   # where the weight goes: blocks by weighted VALU count
        rows = sorted(((w * len(b["ops"]), w, len(b["ops"]), b["label"], kinds.get(comp[i] if comp[i] in cyclic else -1, "-"))
                       for i, (b, w) in enumerate(zip(blocks, weight)) if b["ops"]), reverse=True)
        for r in rows[:40]:
            print("%9.1f = %7.2f x %4d  %-22s %s" % r, file=sys.stderr)
        print("once-per-wave blocks: %.1f VALU in %d blocks" % (sum(len(b["ops"]) for b, w in zip(blocks, weight) if w == 1.0),
                                                                 sum(1 for b, w in zip(blocks, weight) if w == 1.0 and b["ops"])), file=sys.stderr)
    executed = sum(tot.values())
    h_lo = tot["half"] / executed                      # unknown opcodes priced at full rate
    h_hi = (tot["half"] + tot["unknown"]) / executed   # ... at half rate
    hc = os.path.join(ROOT, ".head_commit")   # written beside the snapshot before the GPU call (there is no .git on the box)
    out = {"kernel": ("k_knn_sp<20, true, true, true> (the map's bulk kNN + covariance launch, k = 20, seeded: a re-framed persistent map)" if args.seeded else
                      "k_knn_sp<20, true, true, false> (the map's bulk kNN + covariance launch, k = 20, the full search)"),
           "commit": open(hc).read().strip() if os.path.exists(hc) else None,
           "static_valu_instructions": static, "executed_valu_per_wave": {k: round(v, 1) for k, v in tot.items()},
           "executed_valu_per_query": round(executed / 64.0, 2),
           "trip_counts_per_wave": {k: round(v, 2) for k, v in trips.items()}, "lab": lab,
           "loops_found": {k: sum(1 for x in kinds.values() if x == k) for k in ("scan", "scan_seeded", "drain", "newton", "other")}, "drain_loop_copies": len(drain_blocks),
           "unrolled_newton_steps_found": len(chain),
           "half_rate_fraction": round(h_hi, 4), "half_rate_fraction_if_unmeasured_opcodes_are_full_rate": round(h_lo, 4),
           "peak_full_rate_measured": 1060.0, "peak_half_rate_measured": 595.0,
           "peak_mix_weighted": round(1.0 / (h_hi / 595.0 + (1.0 - h_hi) / 1060.0), 1),
           "peak_mix_weighted_if_unmeasured_opcodes_are_full_rate": round(1.0 / (h_lo / 595.0 + (1.0 - h_lo) / 1060.0), 1),
           "unmeasured_opcodes": {o: round(d[2], 1) for o, d in sorted(per_op.items()) if d[0] == "unknown"},
           "top_opcodes_executed_per_wave": {o: [d[0], round(d[2], 1)] for o, d in sorted(per_op.items(), key=lambda kv: -kv[1][2])[:24]},
           "note": "blocks outside the wave-level loops are weighted 1 per wave (the rare paths -- deferral, exact tie-break, Jacobi set-up -- are a few "
                   "dozen instructions: a slight over-count) and blocks inside a loop by the loop's trip count although a wave skips them when no lane needs them "
                   "(a new piece, a drain): executed_over_pmc is the size of both effects; fp64 transcendental helpers (v_rcp_f64, v_rsq_f64, v_sqrt_f64) are priced at half rate, "
                   "they are slower: the true mix-weighted peak is a little LOWER than printed"}
    if args.pmc and os.path.exists(args.pmc):
        pmc = json.load(open(args.pmc))
        q = pmc.get("valu_wave_instructions_per_query")
        if q:
            out["pmc_valu_per_query"] = q
            out["executed_over_pmc"] = round(executed / 64.0 / q, 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
