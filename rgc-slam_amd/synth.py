"""Synthetic world, local maps, LiDAR scans and trajectories.

Data generation only -- not on the product path.  Everything is analytic (planes, axis-aligned
boxes, vertical cylinders) so rays are cast exactly; all randomness comes from
``numpy.random.default_rng(seed)`` with the seeds recorded by the callers (default 20241008).

The sensor model follows the quantities the reference front-end reconstructs from a cloud
(/root/reference/rgc_slam/src/scanRegistration.cpp:117-213): a clockwise-rotating head
(``ori = -atan2(y, x)`` grows with time), 16 / 64 lasers fired per azimuth step, rel-time from
azimuth, range limits 0.5..80 m (launch/run.launch:12-13) and the self-filter ``x<0 && |y|<0.5``
(scanRegistration.cpp:749).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

SEED = 20241008
GROUND_Z = -0.56  # laderH, scanRegistration.cpp:39

# VLP-16 firing order (elevation in degrees); ring id = int((elev+15)/2+0.5), scanRegistration.cpp:147
VLP16_ELEV = np.array([-15, 1, -13, 3, -11, 5, -9, 7, -7, 9, -5, 11, -3, 13, -1, 15], dtype=np.float64)


def hdl64_elev() -> np.ndarray:
    """64 elevations consistent with the N_SCANS==64 ring formula (scanRegistration.cpp:163-178):
    upper block +2..-8.33 deg (step 1/3), lower block -8.83..-24.33 (step 1/2)."""
    up = 2.0 - np.arange(32) / 3.0
    lo = -8.83 - np.arange(32) * 0.5
    return np.concatenate([up, lo])


def hdl32_elev() -> np.ndarray:
    """32 elevations consistent with the N_SCANS==32 ring formula (scanRegistration.cpp:154-162): scanID = int((angle + 92/3) * 3/4)
    TRUNCATES, so ring k is the band [-92/3 + 4k/3, -92/3 + 4(k+1)/3) degrees; the beams sit in the middle of their bands."""
    return -92.0 / 3.0 + (np.arange(32) + 0.5) * 4.0 / 3.0


@dataclass
class World:
    half_extent: float
    boxes: np.ndarray  # (m, 6): xmin, ymin, zmin, xmax, ymax, zmax
    cyls: np.ndarray   # (c, 4): cx, cy, radius, ztop   (base at GROUND_Z)
    ground_z: float = GROUND_Z
    seed: int = SEED
    ramp: tuple | None = None   # (x0, x1, slope): the ground rises with `slope` between x0 and x1 and stays level behind (a ground CHANGE for
                                # the odometer's ground-change detector, RGC_odometer.cpp:1034-1087); None: the flat ground every config uses

    def ground_height(self, x):
        """ground level at world x (scalar or array)"""
        if self.ramp is None:
            return np.full_like(np.asarray(x, np.float64), self.ground_z)
        x0, x1, s = self.ramp
        return self.ground_z + s * (np.clip(np.asarray(x, np.float64), x0, x1) - x0)


def make_world(half_extent: float = 60.0, seed: int = SEED, pitch: float = 20.0) -> World:
    rng = np.random.default_rng(seed)
    n = int(math.floor(half_extent / pitch))
    boxes, cyls = [], []
    for i in range(-n, n + 1):
        for j in range(-n, n + 1):
            cx = i * pitch + rng.uniform(-4, 4)
            cy = j * pitch + rng.uniform(-4, 4)
            w, d, h = rng.uniform(4, 12), rng.uniform(4, 12), rng.uniform(3, 12)
            kind = rng.uniform()
            if abs(cy) < 6.0 + d / 2:  # keep the +x corridor around the start clear
                continue
            if kind < 0.8:
                boxes.append([cx - w / 2, cy - d / 2, GROUND_Z, cx + w / 2, cy + d / 2, GROUND_Z + h])
            else:
                cyls.append([cx, cy, rng.uniform(0.3, 0.6), GROUND_Z + rng.uniform(4, 8)])
            # a thin pillar next to most lattice sites
            if rng.uniform() < 0.5:
                cyls.append([cx + rng.uniform(-8, 8), cy + np.sign(cy) * rng.uniform(0, 3), rng.uniform(0.15, 0.4),
                             GROUND_Z + rng.uniform(3, 7)])
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 6)
    cyls = np.asarray(cyls, dtype=np.float64).reshape(-1, 4)
    # perimeter walls so that upward beams return something
    L, t, h = half_extent, 0.5, 10.0
    walls = np.array([
        [-L - t, -L - t, GROUND_Z, L + t, -L, GROUND_Z + h],
        [-L - t, L, GROUND_Z, L + t, L + t, GROUND_Z + h],
        [-L - t, -L, GROUND_Z, -L, L, GROUND_Z + h],
        [L, -L, GROUND_Z, L + t, L, GROUND_Z + h],
    ])
    boxes = np.concatenate([boxes, walls], axis=0)
    return World(half_extent=half_extent, boxes=boxes, cyls=cyls, seed=seed)


# ----------------------------------------------------------------------------------------------
# leaf-centroid filter (numpy restatement of pcl::VoxelGrid used only to *build* synthetic maps)
# ----------------------------------------------------------------------------------------------
def leaf_centroids(xyz: np.ndarray, leaf: float) -> np.ndarray:
    ijk = np.floor(xyz / leaf).astype(np.int64)
    ijk -= ijk.min(axis=0)
    dims = ijk.max(axis=0) + 1
    key = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    start = np.flatnonzero(np.concatenate([[True], key_s[1:] != key_s[:-1]]))
    cnt = np.diff(np.concatenate([start, [len(key_s)]]))
    sums = np.add.reduceat(xyz[order], start, axis=0)
    return sums / cnt[:, None]


def _sample_rect(rng, origin, u, v, lu, lv, step, normal, sigma):
    nu, nv = max(int(lu / step), 1), max(int(lv / step), 1)
    a = (np.arange(nu) + 0.5) * (lu / nu)
    b = (np.arange(nv) + 0.5) * (lv / nv)
    A, B = np.meshgrid(a, b, indexing="ij")
    A = A.ravel() + rng.uniform(-0.4, 0.4, A.size) * (lu / nu)
    B = B.ravel() + rng.uniform(-0.4, 0.4, B.size) * (lv / nv)
    p = origin[None, :] + A[:, None] * u[None, :] + B[:, None] * v[None, :]
    p += rng.normal(0.0, sigma, (p.shape[0], 1)) * normal[None, :]
    return p


def make_map(world: World, n_points: int | None, leaf: float = 0.3, seed: int = SEED, sigma: float = 0.005,
             raw_step: float = 0.15) -> np.ndarray:
    """Map cloud = surfaces sampled (noise sigma) then leaf-centroid filtered, trimmed to exactly
    ``n_points`` by keeping the points closest to the origin. Returns float32 (n,3)."""
    rng = np.random.default_rng(seed + 1)
    L = world.half_extent
    ex, ey, ez = np.eye(3)
    parts = [_sample_rect(rng, np.array([-L, -L, world.ground_z]), ex, ey, 2 * L, 2 * L, raw_step, ez, sigma)]
    for bx in world.boxes:
        x0, y0, z0, x1, y1, z1 = bx
        parts.append(_sample_rect(rng, np.array([x0, y0, z0]), ex, ez, x1 - x0, z1 - z0, raw_step, ey, sigma))
        parts.append(_sample_rect(rng, np.array([x0, y1, z0]), ex, ez, x1 - x0, z1 - z0, raw_step, ey, sigma))
        parts.append(_sample_rect(rng, np.array([x0, y0, z0]), ey, ez, y1 - y0, z1 - z0, raw_step, ex, sigma))
        parts.append(_sample_rect(rng, np.array([x1, y0, z0]), ey, ez, y1 - y0, z1 - z0, raw_step, ex, sigma))
        parts.append(_sample_rect(rng, np.array([x0, y0, z1]), ex, ey, x1 - x0, y1 - y0, raw_step, ez, sigma))
    for cx, cy, r, zt in world.cyls:
        h = zt - world.ground_z
        na, nz = max(int(2 * math.pi * r / raw_step), 6), max(int(h / raw_step), 1)
        ang = rng.uniform(0, 2 * math.pi, na * nz)
        zz = world.ground_z + rng.uniform(0, h, na * nz)
        rr = r + rng.normal(0, sigma, na * nz)
        parts.append(np.stack([cx + rr * np.cos(ang), cy + rr * np.sin(ang), zz], axis=1))
    raw = np.concatenate(parts, axis=0)
    cen = leaf_centroids(raw, leaf)
    if n_points is None:
        return np.ascontiguousarray(cen, dtype=np.float32)
    if cen.shape[0] < n_points:
        raise ValueError(f"world too small: {cen.shape[0]} leaf centroids < requested {n_points}")
    # trim the (small) excess uniformly at random so no region of the world loses its map
    keep = np.sort(rng.choice(cen.shape[0], n_points, replace=False))
    return np.ascontiguousarray(cen[keep], dtype=np.float32)


def make_world_and_map(n_points: int, seed: int = SEED, leaf: float = 0.3):
    """World sized so that its leaf-filtered map holds a little over n_points (<= 15 % excess, trimmed
    uniformly at random so no region loses its map).  Returns (world, float32 map (n_points,3))."""
    L = max(12.0, math.sqrt(1.07 * n_points / (100.0 * (0.3 / leaf) ** 2)))
    best = None  # smallest world seen that holds at least n_points (the count moves in steps: whole boxes enter with L)

    def trimmed(world, cen):
        rng = np.random.default_rng(seed + 2)
        keep = np.sort(rng.choice(cen.shape[0], n_points, replace=False))
        return world, np.ascontiguousarray(cen[keep])

    for _ in range(12):
        world = make_world(half_extent=L, seed=seed)
        cen = make_map(world, None, leaf=leaf, seed=seed)
        c = cen.shape[0]
        if n_points <= c <= 1.15 * n_points:
            return trimmed(world, cen)
        if c >= n_points and (best is None or c < best[1].shape[0]):
            best = (world, cen)
        L *= math.sqrt(1.07 * n_points / c)
    if best is not None:  # never inside the 15 % window: take the tightest overshoot and trim more
        return trimmed(*best)
    raise RuntimeError("could not size the world")


# ----------------------------------------------------------------------------------------------
# ray casting
# ----------------------------------------------------------------------------------------------
def _cast(world: World, o: np.ndarray, d: np.ndarray, max_range: float) -> np.ndarray:
    """o, d: (n,3) world-frame origins / unit directions. returns range t (inf = no hit)."""
    n = d.shape[0]
    t_best = np.full(n, np.inf)
    # ground
    if world.ramp is None:
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (world.ground_z - o[:, 2]) / d[:, 2]
        ok = (d[:, 2] < 0) & (tg > 0)
        t_best = np.where(ok, np.minimum(t_best, tg), t_best)
    else:  # three planes -- level, ramp, level -- each valid where its hit falls into its own x range
        x0, x1, sl = world.ramp
        for (a, b, z_at_a, slope) in ((-np.inf, x0, world.ground_z, 0.0), (x0, x1, world.ground_z, sl), (x1, np.inf, world.ground_z + sl * (x1 - x0), 0.0)):
            xa = x0 if not np.isfinite(a) else a
            # plane: z = z_at_a + slope * (x - xa)
            with np.errstate(divide="ignore", invalid="ignore"):
                tg = (z_at_a + slope * (o[:, 0] - xa) - o[:, 2]) / (d[:, 2] - slope * d[:, 0])
                xh = o[:, 0] + tg * d[:, 0]
            ok = np.isfinite(tg) & (tg > 0) & (xh >= a) & (xh <= b)
            t_best = np.where(ok, np.minimum(t_best, tg), t_best)
    # boxes near the sensor only
    oc = o.mean(axis=0)
    if len(world.boxes):
        bx = world.boxes
        near = ((bx[:, 0] < oc[0] + max_range) & (bx[:, 3] > oc[0] - max_range) &
                (bx[:, 1] < oc[1] + max_range) & (bx[:, 4] > oc[1] - max_range))
        bx = bx[near]
        for s in range(0, n, 8192):
            oo, dd = o[s:s + 8192, None, :], d[s:s + 8192, None, :]
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / dd
                t0 = (bx[None, :, 0:3] - oo) * inv
                t1 = (bx[None, :, 3:6] - oo) * inv
            tn = np.nanmax(np.minimum(t0, t1), axis=2)
            tf = np.nanmin(np.maximum(t0, t1), axis=2)
            hit = (tf >= tn) & (tf > 0)
            th = np.where(tn > 1e-9, tn, tf)  # origin inside a box -> exit point
            th = np.where(hit, th, np.inf)
            t_best[s:s + 8192] = np.minimum(t_best[s:s + 8192], th.min(axis=1))
    if len(world.cyls):
        cy = world.cyls
        near = (np.abs(cy[:, 0] - oc[0]) < max_range) & (np.abs(cy[:, 1] - oc[1]) < max_range)
        cy = cy[near]
        for s in range(0, n, 8192):
            oo, dd = o[s:s + 8192], d[s:s + 8192]
            ox = oo[:, None, 0] - cy[None, :, 0]
            oy = oo[:, None, 1] - cy[None, :, 1]
            dx, dy = dd[:, None, 0], dd[:, None, 1]
            a = dx * dx + dy * dy
            b = 2 * (ox * dx + oy * dy)
            c = ox * ox + oy * oy - cy[None, :, 2] ** 2
            disc = b * b - 4 * a * c
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = (-b - np.sqrt(disc)) / (2 * a)
            z = oo[:, None, 2] + tt * dd[:, None, 2]
            ok = (disc > 0) & (tt > 0) & (z >= world.ground_z) & (z <= cy[None, :, 3])
            tt = np.where(ok, tt, np.inf)
            t_best[s:s + 8192] = np.minimum(t_best[s:s + 8192], tt.min(axis=1))
    return t_best


def rot_zyx(yaw: float, pitch: float, roll: float) -> np.ndarray:
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def se3(R: np.ndarray, t) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def make_scan(world: World, T_ws: np.ndarray, elev_deg: np.ndarray = VLP16_ELEV, n_az: int = 1800, seed: int = SEED,
              sigma: float = 0.01, min_range: float = 0.5, max_range: float = 80.0, T_ws_end: np.ndarray | None = None,
              n_points: int | None = None):
    """Cast one sweep.  T_ws: sensor->world pose at sweep start (4x4).  If T_ws_end is given the pose is
    interpolated linearly over the sweep (motion distortion).  Returns dict with float32 ``xyz`` (sensor
    frame at the per-point pose), ``intensity`` (uint8-like float32), ``ring`` (laser id by elevation rank),
    ``rel_time`` in [0,1), in firing (time) order.  If ``n_points`` is set the result is trimmed to exactly
    that many points by dropping random points (order kept)."""
    rng = np.random.default_rng(seed + 7)
    nl = len(elev_deg)
    az_idx = np.repeat(np.arange(n_az), nl)
    las = np.tile(np.arange(nl), n_az)
    rel = (az_idx + las / (nl * 1.0)) / n_az
    ori = 2 * math.pi * rel                  # LOAM's ori = -atan2(y,x) increases with time (clockwise head)
    az = -ori
    el = np.deg2rad(elev_deg)[las]
    d_s = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    if T_ws_end is None:
        R = T_ws[:3, :3]
        o = np.broadcast_to(T_ws[:3, 3], d_s.shape).copy()
        d_w = d_s @ R.T
    else:
        # small-motion interpolation: translation linear, rotation via linear blend + re-orthonormalisation
        t0, t1 = T_ws[:3, 3], T_ws_end[:3, 3]
        o = t0[None, :] + rel[:, None] * (t1 - t0)[None, :]
        Rm = (1 - rel)[:, None, None] * T_ws[:3, :3][None] + rel[:, None, None] * T_ws_end[:3, :3][None]
        U, _, Vt = np.linalg.svd(Rm)
        Rm = U @ Vt
        d_w = np.einsum("nij,nj->ni", Rm, d_s)
    t = _cast(world, o, d_w, max_range)
    t = t + rng.normal(0.0, sigma, t.shape)
    with np.errstate(invalid="ignore"):  # rays that hit nothing carry t = inf; they are dropped by `ok` below
        p = d_s * t[:, None]
    ok = np.isfinite(t) & (t > min_range) & (t < max_range)
    ok &= ~((p[:, 0] < 0) & (np.abs(p[:, 1]) < 0.5))
    # per-surface albedo-like intensity with noise (uint8 range)
    with np.errstate(invalid="ignore"):
        inten = np.clip(40 + 30 * np.sin(0.37 * (o[:, 0] + d_w[:, 0] * t)) + 25 * np.cos(0.23 * (o[:, 1] + d_w[:, 1] * t)) +
                        rng.normal(0, 3, t.shape), 0, 255)
    ring_rank = np.argsort(np.argsort(elev_deg))  # laser id -> ring (ascending elevation)
    out = dict(xyz=p[ok].astype(np.float32), intensity=np.floor(inten[ok]).astype(np.float32),
               ring=ring_rank[las[ok]].astype(np.int32), rel_time=rel[ok].astype(np.float32))
    if n_points is not None:
        m = out["xyz"].shape[0]
        if m < n_points:
            raise ValueError(f"scan has {m} returns < requested {n_points}; raise n_az")
        keep = np.sort(rng.choice(m, n_points, replace=False))
        out = {k: v[keep] for k, v in out.items()}
    return out


def make_scan_n(world: World, T_ws: np.ndarray, n_points: int, elev_deg: np.ndarray = VLP16_ELEV, seed: int = SEED, **kw):
    """Scan with exactly n_points returns: azimuth resolution chosen from the expected return rate."""
    nl = len(elev_deg)
    n_az = int(n_points / nl * 1.15) + 8
    for _ in range(6):
        try:
            return make_scan(world, T_ws, elev_deg, n_az=n_az, seed=seed, n_points=n_points, **kw)
        except ValueError:
            n_az = int(n_az * 1.3)
    raise RuntimeError("could not reach requested scan size")


# ----------------------------------------------------------------------------------------------
# trajectories
# ----------------------------------------------------------------------------------------------
def make_trajectory(n_frames: int, seed: int = SEED, dt: float = 0.1):
    """Constant-twist segments: v in [0.5,2] m/s, yaw-rate in [-0.3,0.3] rad/s, small pitch/roll
    (SURVEY.md §8d).  Returns list of 4x4 sensor->world poses starting at the origin."""
    rng = np.random.default_rng(seed + 3)
    T = np.eye(4)
    poses = [T.copy()]
    v, w = rng.uniform(0.5, 2.0), rng.uniform(-0.3, 0.3)
    for k in range(1, n_frames):
        if k % 25 == 0:
            v, w = rng.uniform(0.5, 2.0), rng.uniform(-0.3, 0.3)
        dR = rot_zyx(w * dt, rng.normal(0, 0.002), rng.normal(0, 0.002))
        dT = se3(dR, [v * dt, rng.normal(0, 0.003), rng.normal(0, 0.001)])
        T = T @ dT
        poses.append(T.copy())
    return poses


def make_imu(poses, dt: float = 0.1, rate: float = 200.0, lead_s: float = 2.5, seed: int = SEED, gyro_sigma: float = 2e-4,
             acc_sigma: float = 5e-3, ba=(0.23054, -0.22046, -0.14313), bg=(0.00127, -0.00061, -0.00267)):
    """sensor_msgs/Imu samples for a trajectory of make_trajectory(): (stamps, acc (n,3), gyr (n,3)), raw -- i.e. INCLUDING the biases
    the odometer subtracts (include/rgc_slam/utility.h:253-254).  Pose k holds at t = k * dt; between poses the body turns at a
    constant rate (the trajectory's constant-twist segments); the accelerometer sees gravity only (specific force of a platform
    that does not accelerate; +z up, /root/reference/rgc_slam/src/RGC_odometer.cpp:598-599 reads roll / pitch off it); the platform
    rests at pose 0 for lead_s seconds first, long enough for the node to drop its first 100 messages and for the attitude filter's
    fast phase (300 samples, :565-572)."""
    rng = np.random.default_rng(seed + 17)
    n_lead = int(round(lead_s * rate))
    n_move = int(round((len(poses) - 1) * dt * rate))
    stamps = (np.arange(-n_lead, n_move + 2) / rate).astype(np.float64)
    acc = np.zeros((len(stamps), 3)); gyr = np.zeros((len(stamps), 3))
    def log_so3(R):
        c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2)); th = math.acos(c)
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        return w * (0.5 if th < 1e-9 else th / (2 * math.sin(th)))
    def exp_so3(w):
        th = np.linalg.norm(w)
        if th < 1e-12:
            return np.eye(3)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K
    for j, t in enumerate(stamps):
        k = int(math.floor(t / dt)) if t > 0 else -1
        if k < 0 or k >= len(poses) - 1:
            R = poses[0][:3, :3] if k < 0 else poses[-1][:3, :3]
            w = np.zeros(3)
        else:
            w = log_so3(poses[k][:3, :3].T @ poses[k + 1][:3, :3]) / dt
            R = poses[k][:3, :3] @ exp_so3(w * (t - k * dt))
        acc[j] = R.T @ np.array([0.0, 0.0, 9.81]) + np.asarray(ba) + rng.normal(0, acc_sigma, 3)
        gyr[j] = w + np.asarray(bg) + rng.normal(0, gyro_sigma, 3)
    return stamps, acc, gyr


def perturb(T: np.ndarray, rng, trans_sigma: float, rot_sigma_rad: float) -> np.ndarray:
    w = rng.normal(0, rot_sigma_rad, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (math.sin(th) / th) * K + ((1 - math.cos(th)) / th ** 2) * K @ K if th > 1e-12 else np.eye(3)
    return se3(R, rng.normal(0, trans_sigma, 3)) @ T
