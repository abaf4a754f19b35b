import sys, os, json, subprocess
for hv in (320, 640, 1280, 2560, 5120):
    env = dict(os.environ, RGC_KNN_HEAVY=str(hv))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "12", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env).stdout
    d = json.loads(out.strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print(hv, d["ms_per_step"], {x: k[x] for x in ("knn_cov_target", "knn_coop_target", "knn_cov_source", "knn_coop_source")}, flush=True)
