"""Design study for the map's bulk kNN launch (round 3): what the LAZY insertion costs a wave, per drain policy.

CPU only (numpy); no product code.  Replays knn_point_sp's selection machinery for sampled waves (64 consecutive queries in cell order)
of the c-main map in lock-step, the way the SIMD runs it: every lane streams its own candidates (nine rows, nearest first, a row whose
bound is not below the lane's tail when the lane reaches it is skipped), four per trip; the first 24 go through the sorting network; a
later candidate whose distance is below the lane's tail AS OF ITS LAST DRAIN is appended to the lane's buffer; a drain runs as many
insert rounds as the policy says, each round popping one key in every lane that has one.  Reported per wave: trips of the scan loop,
insert rounds (what the wave pays: ~31 VALU instructions each), appended keys of the average and of the fullest lane.

    python scripts/sim_drain.py [n_waves]
"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from sim_candidates import build, gaps, ring_order

K = 20
L = K + 2


def lane_stream(g, i):
    """the candidate pieces of query i in visiting order: list of (bound, distances^2 array)"""
    P, start, dim = g["P"], g["start"], g["dim"]
    q = P[i].astype(np.float64)
    c = g["c"][i]
    gap = gaps(g, q, c)
    out = []
    for (dy, dz) in ring_order(1):
        y, z = c[1] + dy, c[2] + dz
        if not (0 <= y < dim[1] and 0 <= z < dim[2]):
            continue
        base = (z * dim[1] + y) * dim[0]
        a, b = start[base + c[0] - 1], start[base + c[0] + 2]
        if b > a:
            out.append((gap(1, dy) ** 2 + gap(2, dz) ** 2, ((P[a:b].astype(np.float64) - q) ** 2).sum(1)))
    return out


class Lane:
    def __init__(self, pieces):
        self.pieces, self.pi, self.off = pieces, 0, 0
        self.chain = np.full(L, np.inf)
        self.tau = np.inf
        self.buf = []
        self.seen = 0
        self.appended = 0

    def live(self):
        return self.pi < len(self.pieces)

    def next_quad(self):
        while self.pi < len(self.pieces):
            bound, d = self.pieces[self.pi]
            if self.off == 0 and bound >= self.tau:   # a piece out of reach is an empty one
                self.pi += 1
                continue
            q = d[self.off:self.off + 4]
            self.off += 4
            if self.off >= len(d):
                self.pi += 1
                self.off = 0
            return q
        return None

    def insert(self, x):
        if x < self.chain[-1]:
            self.chain = np.sort(np.append(self.chain, x))[:L]


def run_wave(g, w, policy, depth=12, trigger=8, low=0):
    lanes = [Lane(lane_stream(g, i)) for i in range(w * 64, w * 64 + 64)]
    # fill: six quads per lane through the sorting network
    for ln in lanes:
        first = []
        for _ in range(6):
            q = ln.next_quad()
            if q is not None:
                first.extend(q.tolist())
        for x in first:
            ln.insert(x)
        ln.tau = ln.chain[-1]
    trips = rounds = 0

    def drain(down_to):
        nonlocal rounds
        while max(len(ln.buf) for ln in lanes) > down_to:
            rounds += 1
            for ln in lanes:
                if ln.buf:
                    ln.insert(ln.buf.pop())
        for ln in lanes:
            ln.tau = ln.chain[-1]

    while any(ln.live() for ln in lanes):
        trips += 1
        for ln in lanes:
            q = ln.next_quad()
            if q is None:
                continue
            for x in q:
                if x < ln.tau:
                    ln.buf.append(x)
                    ln.appended += 1
        mx = max(len(ln.buf) for ln in lanes)
        if policy == "full" and mx > trigger:
            drain(0)
        elif policy == "partial" and mx > trigger:
            drain(low)
        elif policy == "eager":   # one round per trip whenever at least `trigger` lanes hold a key, full drain when a buffer is nearly full
            if mx > depth - 4:
                drain(0)
            elif sum(1 for ln in lanes if ln.buf) >= trigger:
                rounds += 1
                for ln in lanes:
                    if ln.buf:
                        ln.insert(ln.buf.pop())
                    ln.tau = ln.chain[-1]
    drain(0)
    app = np.array([ln.appended for ln in lanes])
    return trips, rounds, app.mean(), app.max()


def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    P = np.load("/tmp/map1m.npy")
    g = build(P, 1.0)
    rng = np.random.default_rng(1)
    waves = rng.integers(0, len(P) // 64, nw)
    print(f"{'policy':46s} {'trips':>6s} {'rounds':>7s} {'appended/lane':>13s} {'fullest lane':>12s}")
    for (name, policy, kw) in [("full drain when a lane holds > 8 (today)", "full", dict(trigger=8)),
                               ("full drain when a lane holds > 4", "full", dict(trigger=4)),
                               ("full drain when a lane holds > 2", "full", dict(trigger=2)),
                               ("drain to 4 when a lane holds > 8", "partial", dict(trigger=8, low=4)),
                               ("drain to 2 when a lane holds > 8", "partial", dict(trigger=8, low=2)),
                               ("drain to 6 when a lane holds > 8", "partial", dict(trigger=8, low=6)),
                               ("drain to 2 when a lane holds > 5", "partial", dict(trigger=5, low=2)),
                               ("one round per trip if >= 16 lanes hold a key", "eager", dict(trigger=16)),
                               ("one round per trip if >= 32 lanes hold a key", "eager", dict(trigger=32))]:
        r = np.array([run_wave(g, int(w), policy, **kw) for w in waves], dtype=np.float64).mean(0)
        print(f"{name:46s} {r[0]:6.1f} {r[1]:7.1f} {r[2]:13.1f} {r[3]:12.1f}", flush=True)


if __name__ == "__main__":
    main()
