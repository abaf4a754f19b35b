"""f3: wire / disk formats.  Host-only pieces (TUM line, .pcd writer, PointField tables) run on the CPU; the PointCloud2
(un)packing kernels are checked on the MI355X against numpy structured dtypes (an independent statement of the byte layout)."""
import numpy as np
import pytest


def test_tum_line_matches_iostream_formatting():
    from rgc_slam_amd import wire
    t, q = [1.23456789012, -0.5, 1e-10], [0.0, 0.1, -0.2, 0.9746794345]
    line = wire.tum_line(1700000000.1234567, t, q)
    assert line == "%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (1700000000.1234567, *t, *q)
    assert line.endswith("\n") and len(line.split()) == 8 and line.split()[0] == "1700000000.123457"


def _read_pcd(path):
    raw = open(path, "rb").read()
    head, _, body = raw.partition(b"DATA ")
    kind, _, body = body.partition(b"\n")
    hdr = dict(l.split(" ", 1) for l in head.decode().strip().split("\n") if not l.startswith("#"))
    n = int(hdr["POINTS"])
    assert hdr["FIELDS"] == "x y z intensity" and hdr["SIZE"] == "4 4 4 4" and hdr["TYPE"] == "F F F F" and hdr["COUNT"] == "1 1 1 1"
    assert hdr["WIDTH"] == str(n) and hdr["HEIGHT"] == "1" and hdr["VERSION"] == "0.7" and hdr["VIEWPOINT"] == "0 0 0 1 0 0 0"
    if kind == b"binary":
        return np.frombuffer(body, np.float32).reshape(n, 4), "binary"
    return np.array([[float(v) for v in l.split()] for l in body.decode().strip().split("\n")], np.float32).reshape(n, 4), "ascii"


def test_pcd_writer_round_trip(tmp_path):
    from rgc_slam_amd import wire
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.normal(0, 30, (500, 3)), rng.uniform(0, 255, (500, 1))], axis=1).astype(np.float32)
    pts[7, 2] = np.nan
    wire.pcd_write(tmp_path / "b.pcd", pts, binary=True)
    got, kind = _read_pcd(tmp_path / "b.pcd")
    assert kind == "binary" and np.array_equal(got, pts, equal_nan=True)
    wire.pcd_write(tmp_path / "a.pcd", pts, binary=False)
    got, kind = _read_pcd(tmp_path / "a.pcd")
    assert kind == "ascii" and np.isnan(got[7, 2])
    ok = ~np.isnan(pts)
    assert np.allclose(got[ok], pts[ok], rtol=1e-7, atol=0)   # 8 significant digits (PCL's default precision): not a bit-exact round trip
    txt = open(tmp_path / "a.pcd").read().split("DATA ascii\n")[1].split("\n")[0]
    assert txt == " ".join("%.8g" % v for v in pts[0])
    wire.pcd_write(tmp_path / "e.pcd", np.zeros((0, 4), np.float32))
    assert _read_pcd(tmp_path / "e.pcd")[0].shape == (0, 4)


def test_point_field_tables():
    from rgc_slam_amd import wire
    f, step = wire.point_fields("xyzi")
    assert step == 32 and f == [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("intensity", 16, 7, 1)]
    f, step = wire.point_fields("xyzinormal")
    assert step == 48 and dict((n, o) for n, o, _, _ in f) == dict(x=0, y=4, z=8, intensity=32, normal_x=16, normal_y=20, normal_z=24, curvature=36)


# ---- device part ------------------------------------------------------------------------------------------------------------
def _msg(n, dtype, seed=1):
    rng = np.random.default_rng(seed)
    a = np.zeros(n, dtype=dtype)
    for name in dtype.names:
        if name.startswith("pad"):
            a[name] = rng.integers(0, 255, a[name].shape)
        elif np.issubdtype(dtype[name], np.floating):
            a[name] = rng.normal(0, 40, n)
        else:
            a[name] = rng.integers(0, np.iinfo(dtype[name]).max // 2, n)
    return a


@pytest.mark.gpu
def test_pc2_unpack_velodyne_layouts():
    from rgc_slam_amd import wire
    w = wire.Wire(0)
    n = 5000
    # the velodyne driver's PointXYZIR message: x y z @0 4 8, intensity @16, ring (uint16) @20, 32-byte points
    dt = np.dtype({"names": ["x", "y", "z", "intensity", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"], "offsets": [0, 4, 8, 16, 20], "itemsize": 32})
    a = _msg(n, dt)
    lay = wire.layout(32, dict(x=(0, 7), y=(4, 7), z=(8, 7), intensity=(16, 7), ring=(20, 4)))
    xyzi, ring, tm = w.unpack(a.tobytes(), n, lay, want_ring=True, want_time=True)
    assert np.array_equal(xyzi, np.stack([a["x"], a["y"], a["z"], a["intensity"]], axis=1))
    assert np.array_equal(ring, a["ring"].astype(np.int32)) and np.all(tm == 0)
    # XYZIRT with a per-point time, 22-byte packed points, and no ring requested
    dt = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                   "offsets": [0, 4, 8, 12, 16, 18], "itemsize": 22})
    a = _msg(n, dt, 2)
    lay = wire.layout(22, dict(x=(0, 7), y=(4, 7), z=(8, 7), intensity=(12, 7), ring=(16, 4), time=(18, 7)))
    xyzi, ring, tm = w.unpack(a.tobytes(), n, lay, want_time=True)
    assert ring is None and np.array_equal(tm, a["time"]) and np.array_equal(xyzi[:, 3], a["intensity"])
    # big-endian doubles for the coordinates, uint8 intensity: converted (strict = False) or left at 0 like fromROSMsg (strict = True)
    dt = np.dtype({"names": ["x", "y", "z", "intensity"], "formats": [">f8", ">f8", ">f8", "u1"], "offsets": [0, 8, 16, 24], "itemsize": 25})
    a = _msg(n, dt, 3)
    lay = wire.layout(25, dict(x=(0, 8), y=(8, 8), z=(16, 8), intensity=(24, 2)), is_bigendian=True)
    xyzi, _, _ = w.unpack(a.tobytes(), n, lay)
    assert np.array_equal(xyzi, np.stack([a["x"], a["y"], a["z"], a["intensity"]], axis=1).astype(np.float32))
    lay = wire.layout(25, dict(x=(0, 8), y=(8, 8), z=(16, 8), intensity=(24, 2)), is_bigendian=True, strict=True)
    xyzi, _, _ = w.unpack(a.tobytes(), n, lay)
    assert np.all(xyzi == 0)      # nothing is FLOAT32: fromROSMsg<PointXYZI> would map no field
    # a message without intensity, and bad layouts
    lay = wire.layout(32, dict(x=(0, 7), y=(4, 7), z=(8, 7)))
    xyzi, _, _ = w.unpack(_msg(n, np.dtype({"names": ["x", "y", "z"], "formats": ["<f4"] * 3, "offsets": [0, 4, 8], "itemsize": 32})).tobytes(), n, lay)
    assert np.all(xyzi[:, 3] == 0)
    from rgc_slam_amd import _lib
    with pytest.raises(_lib.RgcError):
        w.unpack(b"\0" * 64, 2, wire.layout(32, dict(x=(30, 7))))
    w.close()


@pytest.mark.gpu
def test_pc2_pack_matches_pcl_layouts_and_round_trips():
    from rgc_slam_amd import wire
    w = wire.Wire(0)
    rng = np.random.default_rng(4)
    p4 = rng.normal(0, 20, (3000, 4)).astype(np.float32)
    raw = w.pack(p4, "xyzi")
    f, step = wire.point_fields("xyzi")
    dt = np.dtype({"names": ["x", "y", "z", "w", "intensity"], "formats": ["<f4"] * 5, "offsets": [0, 4, 8, 12, 16], "itemsize": step})
    a = np.frombuffer(raw, dt)
    assert np.array_equal(np.stack([a["x"], a["y"], a["z"], a["intensity"]], axis=1), p4) and np.all(a["w"] == 1.0)
    lay = wire.layout(step, {n: (o, t) for n, o, t, _ in f})
    assert np.array_equal(w.unpack(raw, len(p4), lay)[0], p4)
    p5 = rng.normal(0, 20, (2000, 5)).astype(np.float32)
    raw = w.pack(p5, "xyzinormal")
    f, step = wire.point_fields("xyzinormal")
    dt = np.dtype({"names": [n for n, _, _, _ in f], "formats": ["<f4"] * len(f), "offsets": [o for _, o, _, _ in f], "itemsize": step})
    a = np.frombuffer(raw, dt)
    assert np.array_equal(np.stack([a["x"], a["y"], a["z"], a["intensity"], a["normal_x"]], axis=1), p5)
    assert np.all(a["normal_y"] == 0) and np.all(a["curvature"] == 0)
    w.close()


@pytest.mark.gpu
def test_frontend_on_message_bytes_matches_host_path():
    """PointCloud2 bytes -> unpack kernel -> front-end, all on the device, equals the front-end on the host array"""
    import rgc_slam_amd.synth as synth
    from rgc_slam_amd import wire, frontend
    w_ = synth.make_world(seed=synth.SEED)
    sc = synth.make_scan(w_, np.eye(4), n_az=900, seed=synth.SEED + 3)
    xyzi = np.concatenate([sc["xyz"], sc["intensity"][:, None]], axis=1).astype(np.float32)
    dt = np.dtype({"names": ["x", "y", "z", "intensity", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"], "offsets": [0, 4, 8, 16, 20], "itemsize": 32})
    msg = np.zeros(len(xyzi), dt)
    msg["x"], msg["y"], msg["z"], msg["intensity"] = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2], xyzi[:, 3]
    lay = wire.layout(32, dict(x=(0, 7), y=(4, 7), z=(8, 7), intensity=(16, 7), ring=(20, 4)), strict=True)
    fe = frontend.ScanRegistration(device=0)
    a = fe.laserCloudHandler(xyzi)
    b = fe.laserCloudHandlerMsg(msg.tobytes(), len(xyzi), lay)
    for k in ("cloud", "sharp", "flat", "inten", "label", "curvature", "groundparam"):
        assert np.array_equal(a[k], b[k]), k
