"""Randomised campaign of the PointCloud2 unpacking (f3) against numpy's structured dtypes: random point steps, field orders and offsets (packed,
padded, unaligned), every PointField datatype for every field, both byte orders, fields missing, strict (pcl::fromROSMsg: only FLOAT32 maps)
or converting; and pack -> unpack round trips.      python tests/fuzz/fuzz_wire.py [trials] [seed]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from rgc_slam_amd import wire, _lib

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NP = {1: "i1", 2: "u1", 3: "i2", 4: "u2", 5: "i4", 6: "u4", 7: "f4", 8: "f8"}
SIZE = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 4, 8: 8}
rep = {"trials": 0, "unpacked": 0, "refused_as_expected": 0, "round_trips": 0, "failures": []}
w = wire.Wire(0)
t0 = time.time()
for trial in range(trials):
    rng = np.random.default_rng(seed0 * 2750159 + trial)
    n = int(rng.integers(1, 20000))
    big = bool(rng.random() < 0.3)
    strict = bool(rng.random() < 0.3)
    names = [f for f in wire.FIELDS if f in ("x", "y", "z") or rng.random() < 0.7]
    order = list(rng.permutation(len(names)))
    fields, off = {}, int(rng.integers(0, 5))
    for j in order:
        ty = 7 if (names[j] in ("x", "y", "z") and rng.random() < 0.7) else int(rng.integers(1, 9))
        fields[names[j]] = (off, ty)
        off += SIZE[ty] + int(rng.integers(0, 6))
    step = off + int(rng.integers(0, 9))
    tag = {"trial": trial, "n": n, "big": big, "strict": strict, "step": step, "fields": {k: list(v) for k, v in fields.items()}}
    try:
        dt = np.dtype({"names": list(fields), "formats": [(">" if big else "<") + NP[fields[k][1]] for k in fields], "offsets": [fields[k][0] for k in fields], "itemsize": step})
        a = np.zeros(n, dt)
        for k in fields:
            ty = fields[k][1]
            if ty >= 7:
                a[k] = rng.normal(0, 30, n).astype(NP[ty])
            else:
                info = np.iinfo(NP[ty])
                a[k] = rng.integers(max(info.min, -30000), min(info.max, 60000), n, endpoint=True).astype(NP[ty])
        lay = wire.layout(step, fields, is_bigendian=big, strict=strict)
        xyzi, ring, tm = w.unpack(a.tobytes(), n, lay, want_ring=True, want_time=True)
        rep["unpacked"] += 1

        def col(name, out_dtype):
            if name not in fields:
                return np.full(n, -1 if name == "ring" else 0, out_dtype)   # (rgc_hip.h: a field the message does not have -> 0, ring -> -1)
            ty = fields[name][1]
            if strict and name in ("x", "y", "z", "intensity") and ty != 7:   # pcl::fromROSMsg<PointXYZI> maps a field only when it is FLOAT32
                return np.zeros(n, out_dtype)                                   # (ring and time are not PointXYZI's: converted either way)
            return a[name].astype(out_dtype)
        exp = np.stack([col("x", np.float32), col("y", np.float32), col("z", np.float32), col("intensity", np.float32)], axis=1)
        if not (np.array_equal(xyzi, exp) and np.array_equal(ring, col("ring", np.int32)) and np.array_equal(tm, col("time", np.float32))):
            which = [nm for nm, g, e in (("xyzi", xyzi, exp), ("ring", ring, col("ring", np.int32)), ("time", tm, col("time", np.float32))) if not np.array_equal(g, e)]
            rep["failures"].append(dict(tag, error="unpacked values differ", which=which))
        if trial % 5 == 0:   # pack -> unpack
            kind = "xyzi" if rng.random() < 0.5 else "xyzinormal"
            pts = rng.normal(0, 30, (n, 4 if kind == "xyzi" else 5)).astype(np.float32)
            msg = w.pack(pts, kind)
            stepk = 32 if kind == "xyzi" else 48
            layk = wire.layout(stepk, dict(x=(0, 7), y=(4, 7), z=(8, 7), intensity=(16 if kind == "xyzi" else 32, 7)))
            back, _, _ = w.unpack(msg, n, layk)
            rep["round_trips"] += 1
            if not np.array_equal(back, pts[:, :4]):
                rep["failures"].append(dict(tag, error="pack -> unpack", kind=kind))
    except _lib.RgcError as e:
        rep["failures"].append(dict(tag, error="refused: %s" % (str(e)[:200],)))
    except Exception as e:
        rep["failures"].append(dict(tag, error="exception: %r" % (e,)))
    rep["trials"] += 1
    if len(rep["failures"]) > 15:
        break
# layouts that must be refused: a field that sticks out of the point, a datatype that does not exist, a point step of zero
for bad in (wire.layout(16, dict(x=(14, 7))), wire.layout(16, dict(x=(0, 9), y=(4, 7), z=(8, 7))), wire.layout(0, dict(x=(0, 7)))):
    try:
        w.unpack(b"\0" * 64, 2, bad)
        rep["failures"].append(dict(error="a bad layout was accepted", step=int(bad.point_step)))
    except (_lib.RgcError, ValueError, ZeroDivisionError):
        rep["refused_as_expected"] += 1
w.close()
rep["wall_s"] = round(time.time() - t0, 1)
print(json.dumps(rep))
