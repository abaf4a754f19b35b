"""TEST INFRASTRUCTURE -- a second, deliberately literal restatement of the per-point stages of the scan front-end
(/root/reference/rgc_slam/src/scanRegistration.cpp), written from the reference text line by line in numpy float32 / Python
scalars, to pin oracle/rgc_oracle_aux.c (orc_frontend): A3 range / incidence / near-intensity smoothing (:234-268), A4 curvature
stencils (:270-306), A6 occlusion mask (:433-456), A7 per-sector sort and greedy selection (:469-644) and the intensity append of
:645-656, and (round 6) A5, the ground marking and the weighted plane fit (:308-431, `ground` below) and A2, the ring bucket (:116-230,
`ring_bucket` below; also pinned by the sensor-model properties in tests/test_oracle_frontend.py).  Plain loops: use on small sweeps only.  Per-frame arrays start at zero and
std::sort's unspecified tie order is fixed as ascending index (item 10), like every other implementation in this repository.
PARITY UNPINNED: the reference holds no vectors for this stage and cannot be built here; this is the builder's second restatement.
"""
import numpy as np

f32 = np.float32


def stencils(cloud_xyz, intensity_num_in):
    """:234-306.  cloud_xyz: (n,3) float32 ring-major; intensity_num_in: (n,) ints (deque<int> intensity_num2).  Returns a dict."""
    P = np.asarray(cloud_xyz, np.float32)
    n = len(P)
    num2 = [int(v) for v in intensity_num_in]
    num = list(num2)
    rng = np.zeros(n, np.float32)
    ang = np.zeros(n, np.float32)
    for i in range(n):  # :235-238 float expression, sqrt of a float
        x, y, z = P[i]
        rng[i] = np.sqrt(f32(f32(f32(x * x) + f32(y * y)) + f32(z * z)))
    for i in range(5, n - 5):  # :240-255, Eigen::Vector3d arithmetic
        if rng[i] < 2:
            a, b, now = P[i + 5].astype(np.float64), P[i - 5].astype(np.float64), P[i].astype(np.float64)
            c = (a + b) / 2
            nrm = np.cross(a - b, now - c)
            v = f32(np.dot(nrm, now) / (np.linalg.norm(nrm) * np.linalg.norm(now)))
            ang[i] = -v if v < 0 else v
    for i in range(5, n - 5):  # :257-268, every store truncates to int
        if ang[i] < 0.07 and rng[i] < 2:
            num[i] = int(0.9 * num2[i])
            for j in range(-5, 6):
                if j != 0:
                    num[i] = int(num[i] + 0.005 * num2[i + j])
    curv, curv2, icurv = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    dsrc, osrc = np.zeros(n, np.float32), np.zeros(n, np.float32)

    def lap(col, i):  # p[i-5] + ... + p[i-1] - 10 p[i] + p[i+1] + ... + p[i+5], float, left to right
        s = col[i - 5]
        for k in (-4, -3, -2, -1):
            s = f32(s + col[i + k])
        s = f32(s - f32(f32(10) * col[i]))
        for k in (1, 2, 3, 4, 5):
            s = f32(s + col[i + k])
        return s
    for i in range(5, n - 5):
        dx, dy, dz = lap(P[:, 0], i), lap(P[:, 1], i), lap(P[:, 2], i)
        di = f32(num[i - 5] + num[i - 4] + num[i - 3] + num[i - 2] + num[i - 1] - 10 * num[i] + num[i + 1] + num[i + 2] + num[i + 3] + num[i + 4] + num[i + 5])
        dis = f32(2.0 / (1.0 + float(rng[i]) / 20.0))  # double expression stored in a float
        if dis < 0.2:
            dis = f32(0.2)
        curv[i] = f32(f32(f32(f32(dx * dx) + f32(dy * dy)) + f32(dz * dz)) * dis)
        dsrc[i] = f32(0.5 + float(dis))
        if ang[i] < 0.07 and rng[i] < 2:
            osrc[i] = f32(float(f32(ang[i] * f32(10))) + 0.6)
            icurv[i] = f32((float(ang[i]) + 0.3) * float(di))
        else:
            osrc[i] = f32(3)
            icurv[i] = di
        # float sum of the first five, then double from "- 10.0 * range" on; stored in a float
        s = rng[i - 5]
        for k in (-4, -3, -2, -1):
            s = f32(s + rng[i + k])
        d = float(s) - 10.0 * float(rng[i])
        for k in (1, 2, 3, 4, 5):
            d = d + float(rng[i + k])
        dr = f32(d)
        curv2[i] = abs(f32(dr * dis))
    return dict(range=rng, angle=ang, intensity_num=np.array(num, np.int64), curvature=curv, curvature2=curv2, inten_curvature=icurv,
                distance_source=dsrc, other_source=osrc)


def occlusion(rng):
    """:433-456 on zeroed cloudNeighborPicked"""
    n = len(rng)
    picked = np.zeros(n + 8, np.int32)
    for i in range(5, n - 5):
        d1, d2 = rng[i], rng[i + 1]
        if f32(d1 - d2) > 0.04 * float(d2):
            picked[i - 5:i + 1] = 1
        elif f32(d2 - d1) > 0.04 * float(d1):
            picked[i + 1:i + 7] = 1
    return picked[:n]


def select(cloud, st, picked_in, ground_marked, scan_start, scan_end, use_intensity=1):
    """:458-656.  cloud: (n,4) ring-major with encoded intensity; st: stencils() output; scan_start / scan_end = scanStartInd / scanEndInd."""
    P = np.asarray(cloud, np.float32)
    n = len(P)
    curv, curv2, icurv, num = st["curvature"], st["curvature2"], st["inten_curvature"], st["intensity_num"]
    picked = np.array(picked_in, np.int32).copy()
    ipicked = np.zeros(n, np.int32)
    label, ilabel = np.zeros(n, np.int32), np.zeros(n, np.int32)
    sharp, flat, inten = [], [], []

    def gap(a, b):
        d = P[a, :3] - P[b, :3]
        return f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))

    def suppress(ind, flags, far):
        for l in range(1, 6):
            if far(ind + l, ind + l - 1):
                break
            flags[ind + l] = 1
        for l in range(-1, -6, -1):
            if far(ind + l, ind + l + 1):
                break
            flags[ind + l] = 1
    far_pts = lambda a, b: gap(a, b) > 0.05
    far_int = lambda a, b: abs(f32(int(num[a]) - int(num[b]))) > 35
    for i in range(len(scan_start)):
        S, E = int(scan_start[i]), int(scan_end[i])
        if E - S < 10:
            continue
        for j in range(6):
            sp = S + (E - S) * j // 6
            ep = S + (E - S) * (j + 1) // 6 - 1
            by_curv = sorted(range(sp, ep + 1), key=lambda t: (curv[t], t))
            by_int = sorted(range(sp, ep + 1), key=lambda t: (icurv[t], t))
            k = 0
            for ind in reversed(by_curv):
                if picked[ind] == 0 and ground_marked[ind] != 1 and curv[ind] > 0.1 and curv2[ind] > 0.3:
                    k += 1
                    if k <= 20:
                        label[ind] = 2
                        sharp.append((P[ind, 0], P[ind, 1], P[ind, 2], P[ind, 3], f32(st["distance_source"][ind] + f32(1))))
                    elif k <= 21:
                        label[ind] = 1
                    else:
                        break
                    picked[ind] = 1
                    suppress(ind, picked, far_pts)
            k = 0
            for ind in by_curv:
                if picked[ind] == 0 and curv[ind] < 0.3 and curv2[ind] < 0.4:
                    k += 1
                    if k <= 40:
                        label[ind] = -1
                        flat.append((P[ind, 0], P[ind, 1], P[ind, 2], P[ind, 3], st["distance_source"][ind]))
                    else:
                        break
                    picked[ind] = 1
                    suppress(ind, picked, far_pts)
            k = 0
            for ind in reversed(by_int):
                if ipicked[ind] == 0 and ground_marked[ind] != 1 and icurv[ind] > 65 and label[ind] != 2 and label[ind] != 1:
                    k += 1
                    if k <= 20:
                        ilabel[ind] = 2
                        inten.append((P[ind, 0], P[ind, 1], P[ind, 2], P[ind, 3], st["other_source"][ind]))
                    elif k <= 21:
                        ilabel[ind] = 1
                    else:
                        break
                    ipicked[ind] = 1
                    suppress(ind, ipicked, far_int)
    n_sharp_own = len(sharp)
    if use_intensity and len(flat) and n_sharp_own / len(flat) < 0.3:   # :645-656
        sharp = sharp + inten
    arr = lambda v: np.array(v, np.float32).reshape(-1, 5)
    return dict(label=label, inten_label=ilabel, picked=picked, ipicked=ipicked, sharp=arr(sharp), flat=arr(flat), inten=arr(inten),
                n_sharp_own=n_sharp_own)


def _libm():
    """glibc's single-precision atan2f / atanf / sqrtf: what `atan2(float, float)`, `atan(float)` and `sqrt(float)` resolve to in the reference's C++
    (libstdc++'s <cmath> float overloads) on its x86-64 build -- numpy's float32 arctan2 is another implementation and may differ in the last bit"""
    import ctypes as C
    m = C.CDLL("libm.so.6")
    for f in (m.atan2f, m.atanf, m.sqrtf):
        f.restype = C.c_float
    m.atan2f.argtypes = [C.c_float, C.c_float]
    m.atanf.argtypes = [C.c_float]
    m.sqrtf.argtypes = [C.c_float]
    return m


def ring_bucket(raw_xyzi, n_scans=16, scan_period=0.1):
    """A2, :116-230 restated statement by statement: start / end orientation, per point the vertical angle -> scanID (the three N_SCANS formulas,
    their mixed float / double arithmetic as C++ promotes it), the orientation with the halfPassed logic, relTime, intensity = scanID + scanPeriod *
    relTime, the push into laserCloudScans[scanID] / intensityScans[scanID]; then the concatenation in ring order with scanStartInd / scanEndInd.
    raw_xyzi: (n, 4) float32 AFTER the A1 filter (removeNaN + removeClosedPointCloud).  Returns dict(cloud (m, 4) float32, intensity_num (m,) int64 --
    `int point_intensity = ...intensity` truncates --, ring_count, scan_start, scan_end).  Plain loop: small sweeps."""
    import math
    m = _libm()
    P = np.asarray(raw_xyzi, np.float32)
    n = len(P)
    PI = math.pi
    startOri = f32(-m.atan2f(float(P[0, 1]), float(P[0, 0])))                                   # :117
    endOri = f32(float(f32(-m.atan2f(float(P[n - 1, 1]), float(P[n - 1, 0])))) + 2 * PI)        # :118 (float + double -> float)
    if float(f32(endOri - startOri)) > 3 * PI:                                                   # :120-127
        endOri = f32(float(endOri) - 2 * PI)
    elif float(f32(endOri - startOri)) < PI:
        endOri = f32(float(endOri) + 2 * PI)
    halfPassed = False
    scans = [[] for _ in range(n_scans)]
    inten = [[] for _ in range(n_scans)]
    for i in range(n):
        x, y, z = P[i, 0], P[i, 1], P[i, 2]
        point_intensity = int(P[i, 3])                                                           # :140 (float -> int truncates)
        hyp = f32(m.sqrtf(float(f32(f32(x * x) + f32(y * y)))))                                  # :142: sqrt(x * x + y * y) in float
        va = f32(float(f32(m.atanf(float(f32(z / hyp))))) * 180 / PI)                            #       atan(float) in float, * 180 / M_PI in double, stored float
        if n_scans == 16:
            scanID = int(float(f32(f32(va + f32(15)) / f32(2))) + 0.5)                           # :147 (float, float, then + 0.5 in double)
            if scanID > n_scans - 1 or scanID < 0:
                continue
        elif n_scans == 32:
            scanID = int((float(va) + 92.0 / 3.0) * 3.0 / 4.0)                                   # :156 (all double)
            if scanID > n_scans - 1 or scanID < 0:
                continue
        elif n_scans == 64:
            if float(va) >= -8.83:                                                               # :165-172
                scanID = int(float(f32(f32(2) - va)) * 3.0 + 0.5)
            else:
                scanID = n_scans // 2 + int((-8.83 - float(va)) * 2.0 + 0.5)
            if float(va) > 2 or float(va) < -24.33 or scanID > 50 or scanID < 0:
                continue
        else:
            raise ValueError("wrong scan number")
        ori = f32(-m.atan2f(float(y), float(x)))                                                 # :187
        if not halfPassed:
            if float(ori) < float(startOri) - PI / 2:
                ori = f32(float(ori) + 2 * PI)
            elif float(ori) > float(startOri) + PI * 3 / 2:
                ori = f32(float(ori) - 2 * PI)
            if float(f32(ori - startOri)) > PI:
                halfPassed = True
        else:
            ori = f32(float(ori) + 2 * PI)
            if float(ori) < float(endOri) - PI * 3 / 2:
                ori = f32(float(ori) + 2 * PI)
            elif float(ori) > float(endOri) + PI / 2:
                ori = f32(float(ori) - 2 * PI)
        relTime = f32(f32(ori - startOri) / f32(endOri - startOri))                              # :207
        scans[scanID].append((x, y, z, f32(scanID + scan_period * float(relTime))))              # :210 (int + double * float -> float)
        inten[scanID].append(point_intensity)
    cloud, inum, scan_start, scan_end = [], [], [], []
    for r in range(n_scans):                                                                     # :222-230
        scan_start.append(len(cloud) + 5)
        cloud += scans[r]
        inum += inten[r]
        scan_end.append(len(cloud) - 5)
    return dict(cloud=np.array(cloud, np.float32).reshape(-1, 4), intensity_num=np.array(inum, np.int64), ring_count=np.array([len(q) for q in scans], np.int32),
                scan_start=np.array(scan_start, np.int32), scan_end=np.array(scan_end, np.int32))


GROUND_SCAN_IND = 7            # groundScanInd, :34
LADER_H = 0.56                 # laderH, :39
GROUND_SCAN_RANGE = np.array([2.66, 3.04, 3.56, 4.30, 5.44, 7.41, 11.63, 27.12] + [0.0] * 8, np.float32)   # Ground_scan_range[16], :40


def ground(cloud, ring_count, rng):
    """:308-431 restated line by line.  cloud: (n, >=3) float32, ring-major; ring_count[i] = laserCloudScans[i].points.size(); rng =
    range_vec (stencils()["range"]).  Returns (groundcloudMarked, GroundPoints indices in push order, groundparam[11] or None).
    Reference quirks kept: `i / (groundScanInd - 1)` is an INTEGER division (:323, :325: threshold 0.8 and weight 1.5 for rings 0-5,
    1.6 and 0.5 for ring 6); the neighbour growth runs n = -5 .. 4 (:333); a ring of fewer than 11 points is skipped (the reference's
    size_t `size() - 5` would wrap: undefined there)."""
    P = np.asarray(cloud, np.float32)
    n = len(P)
    mark = np.zeros(n, np.int32)
    pushed, weights = [], []
    center, gweights = np.zeros(3), 0.0
    start = 0
    for i in range(min(GROUND_SCAN_IND, len(ring_count))):
        sz = int(ring_count[i])
        if sz >= 11:
            th = f32(0.8 * (1.0 + i // (GROUND_SCAN_IND - 1)))
            w = 1.5 - i // (GROUND_SCAN_IND - 1)
            for col in range(5, sz - 5):
                ci = start + col
                diff = f32(abs(f32(rng[ci] - GROUND_SCAN_RANGE[i])))
                if diff < th and P[ci, 2] < 0.3:
                    mark[ci] = 1
                    for nn in range(-5, 5):
                        if f32(abs(f32(rng[ci + nn] - rng[ci]))) < f32(th / f32(2)):
                            mark[ci + nn] = 1
                            pushed.append(ci + nn)
                            weights.append(w)
                            center = center + w * P[ci + nn, :3].astype(np.float64)
                            gweights = gweights + w
        start += sz
    if not pushed:
        return mark, np.zeros(0, np.int64), None
    near = P[np.asarray(pushed), :3].astype(np.float64)
    lw = np.asarray(weights)
    center = center / gweights
    cov = np.zeros((3, 3))
    for j in range(len(near)):
        d = near[j] - center
        cov = cov + lw[j] * np.outer(d, d)
    cov = cov / gweights
    ev, V = np.linalg.eigh(cov)                      # ascending, like SelfAdjointEigenSolver
    nrm = V[:, 0] / np.linalg.norm(V[:, 0])
    if center @ nrm < 0:
        nrm = -nrm
    distance, src1 = 0.0, 0.0
    for j in range(len(near)):
        d = near[j] - center
        dl = np.linalg.norm(d)
        dw = 1.0 if dl == 0 else 1.0 - 100.0 * abs(nrm @ (d / dl))   # (Eigen's normalized() of a zero vector stays zero)
        if dw < 0:
            dw = 0.1
        src1 += dw
        distance += dw * (nrm @ near[j])
    distance = distance / src1
    src1 = src1 / len(near)
    if distance / LADER_H > 1.1 or distance / LADER_H < 0.9:
        distance = LADER_H
    if src1 < 0.9:
        distance = 0.9 * LADER_H + 0.1 * distance
    return mark, np.asarray(pushed, np.int64), np.array([*nrm, *V[:, 1], *V[:, 2], distance, 1.0 - src1])
