"""Design study for the map's bulk kNN launch (round 3): how many candidates does a query look at, and how many of them enter
the sorted chain, per search layout?

CPU only (numpy); no product code.  For a sample of waves (64 consecutive queries in cell order) of the c-main map it replays the
lane-per-query search the way the kernel walks it -- pieces nearest first, quads of four candidates, the first 24 candidates through
the sorting network, then the (k+2)-th smallest distance so far as the bound a candidate must beat to be appended -- and reports, per
query and per wave (a wave runs as long as its longest lane): candidates, quads, pieces entered, appended keys.

    python scripts/sim_candidates.py [n_waves]
"""
import sys
import numpy as np

K = 20
L = K + 2


def build(P, res):
    c = np.floor(P.astype(np.float64) / res - 0.5).astype(np.int64)
    mn = c.min(0) - 2
    c -= mn
    dim = c.max(0) + 3
    idx = (c[:, 2] * dim[1] + c[:, 1]) * dim[0] + c[:, 0]
    order = np.argsort(idx, kind="stable")
    cnt = np.bincount(idx, minlength=int(dim.prod()) + 1)
    start = np.concatenate([[0], np.cumsum(cnt)])
    return dict(res=res, mn=mn, dim=dim, start=start, P=P[order], c=c[order])


def gaps(g, q, c):
    res = g["res"]
    wall = (c + g["mn"] + 0.5) * res
    wlo = np.maximum(q - wall, 0.0)
    whi = np.maximum(wall + res - q, 0.0)

    def gap(a, d):
        return 0.0 if d == 0 else (wlo[a] + (-d - 1) * res if d < 0 else whi[a] + (d - 1) * res)
    return gap


def ring_order(R):
    rows = [(max(abs(dy), abs(dz)), dy * dy + dz * dz, dy, dz) for dz in range(-R, R + 1) for dy in range(-R, R + 1)]
    rows.sort()
    return [(dy, dz) for (_, _, dy, dz) in rows]


def pieces_rows(R, clip):
    """today's layout: whole rows of 2R+1 cells, nearest first; clip = cut each row in x at cell granularity by the current bound"""
    return [("row", dy, dz, -R, R, clip) for (dy, dz) in ring_order(R)]


def pieces_two_phase():
    """inner 3x3x3 block first (nine rows of three cells), then the shell: the inner rows' cells -2 / +2, the sixteen outer rows"""
    out = [("row", dy, dz, -1, 1, False) for (dy, dz) in ring_order(1)]
    for (dy, dz) in ring_order(1):
        out.append(("row", dy, dz, -2, -2, False))
        out.append(("row", dy, dz, 2, 2, False))
    for (dy, dz) in ring_order(2):
        if max(abs(dy), abs(dz)) == 2:
            out.append(("row", dy, dz, -2, 2, True))
    return out


def simulate(g, i, pieces, skip, tau0_from_block=None):
    P, start, dim, res = g["P"], g["start"], g["dim"], g["res"]
    q = P[i].astype(np.float64)
    c = g["c"][i]
    gap = gaps(g, q, c)
    best = np.full(L, np.inf)
    ncand = nquad = npiece = nins = seen = 0
    for (_, dy, dz, xa, xb, clip) in pieces:
        tau = best[-1]
        y, z = c[1] + dy, c[2] + dz
        if not (0 <= y < dim[1] and 0 <= z < dim[2]):
            continue
        gx = 0.0 if xa <= 0 <= xb else min(gap(0, xa), gap(0, xb))
        g2 = gap(1, dy) ** 2 + gap(2, dz) ** 2
        if skip and g2 + gx * gx >= tau:
            continue
        if clip and np.isfinite(tau):
            rem = tau - g2
            while xa < 0 and gap(0, xa) ** 2 >= rem:
                xa += 1
            while xb > 0 and gap(0, xb) ** 2 >= rem:
                xb -= 1
        base = (z * dim[1] + y) * dim[0]
        a, b = start[base + c[0] + xa], start[base + c[0] + xb + 1]
        if b <= a:
            continue
        npiece += 1
        d2 = ((P[a:b].astype(np.float64) - q) ** 2).sum(1)
        ncand += b - a
        nquad += (b - a + 3) // 4
        for d in d2:  # appended if it beats the bound (the first 24 go through the sorting network instead)
            seen += 1
            if seen > 24 and d < best[-1]:
                nins += 1
            if d < best[-1]:
                best = np.sort(np.append(best, d))[:L]
    return ncand, nquad, npiece, nins


def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    P = np.load("/tmp/map1m.npy")
    rng = np.random.default_rng(1)
    print(f"{'layout':44s} {'cand':>6s} {'quads':>6s} {'pieces':>6s} {'appends':>7s} | per wave (max lane): quads pieces appends")
    for (res, pieces, skip, name) in [(1.0, pieces_rows(1, False), True, "1.0 m, 3x3 rows, skip (today)"),
                                      (0.5, pieces_rows(2, False), True, "0.5 m, 5x5 rows, skip"),
                                      (0.5, pieces_rows(2, True), True, "0.5 m, 5x5 rows, skip + x clip"),
                                      (0.5, pieces_two_phase(), True, "0.5 m, inner block first, then clipped shell")]:
        g = build(P, res)
        n = len(P)
        waves = rng.integers(0, n // 64, nw)
        tot = np.zeros(4)
        wmax = np.zeros(3)
        for w in waves:
            r = np.array([simulate(g, i, pieces, skip) for i in range(w * 64, w * 64 + 64)], dtype=np.float64)
            tot += r.sum(0)
            wmax += r[:, 1:].max(0)
        tot /= nw * 64
        wmax /= nw
        print(f"{name:44s} {tot[0]:6.1f} {tot[1]:6.1f} {tot[2]:6.1f} {tot[3]:7.1f} | {wmax[0]:6.1f} {wmax[1]:6.1f} {wmax[2]:6.1f}")


if __name__ == "__main__":
    main()
