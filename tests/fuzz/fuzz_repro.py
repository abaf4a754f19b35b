"""Reproduce one trial of tests/fuzz/fuzz_modes.py and say where two routes differ.   python tests/fuzz/fuzz_repro.py <trial> [seed]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rgc_slam_amd.synth as synth
from rgc_slam_amd import registration
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
trial = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1; nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
import importlib.util
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_modes.py")).read()
ns = {}
exec("import numpy as np\nimport rgc_slam_amd.synth as synth\n" + "def cloud" + src.split("def cloud")[1].split("KINDS = ")[0], ns)
KINDS = ["synth", "synth", "uniform", "lattice", "sheets", "clump", "repeats"]
rng = np.random.default_rng(seed0 * 100003 + trial)
kind = KINDS[int(rng.integers(0, len(KINDS)))]
n = int(np.exp(rng.uniform(np.log(2000), np.log(nmax))))
res = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
k = int(rng.choice([20, 20, 20, 20, 10, 25]))
pts = ns["cloud"](kind, n, rng); n = len(pts)
a = np.zeros((n, 4), np.float32); a[:, :3] = pts
print(kind, n, res, k)
def mk(mode):
    v = registration.odometer_vgicp(0); v.setResolution(res); v.setCorrespondenceRandomness(k); v.setNeighbourReuse(mode); return v
A, B = mk(0), mk(0)
dmA, dbA, dmB, dbB = A.device_alloc(a.nbytes), A.device_alloc(a.nbytes), B.device_alloc(a.nbytes), B.device_alloc(a.nbytes)
A.upload(dmA, a); B.upload(dmB, a)
ang = rng.uniform(-np.pi, np.pi, 3) * np.array([1.0, 0.03, 0.03])
Tw = synth.se3(synth.rot_zyx(*ang), rng.uniform(-60, 60, 3) * np.array([1, 1, 0.05]))
q, t = bench.world_to_body(Tw)
A.setInputTargetReframed(dmA, n, 16, q, t, dbA)
B.transformCloudDevice(dmB, n, 16, q, t, dbB); B.setInputTargetDevice(dbB, n, 16)
ba, bb = A.download(dbA, (n, 4)), B.download(dbB, (n, 4))
print("bodies equal:", np.array_equal(ba, bb))
ca, cb = A.getTargetCovariances().reshape(n, -1), B.getTargetCovariances().reshape(n, -1)
d = np.nonzero(np.any(ca != cb, axis=1))[0]
print("points that differ:", len(d), "max |diff|", float(np.abs(ca - cb).max()))
print("stats A", A.stats()); print("stats B", B.stats())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle as orc
oc, _ = orc.covariances(ba[:, :3].copy(), k=k, threads=14); oc = oc.reshape(n, -1)
print("A vs oracle max", float(np.abs(ca - oc).max()), " B vs oracle max", float(np.abs(cb - oc).max()))
for i in d[:6]:
    print(i, ba[i, :3], "\n  A", ca[i], "\n  B", cb[i], "\n  O", oc[i])
idx, _ = orc.knn(ba[:, :3].copy(), k=k, threads=14) if hasattr(orc, "knn") else (None, None)
