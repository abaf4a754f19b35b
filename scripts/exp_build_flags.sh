#!/bin/bash
# The A/B routes that were environment knobs until round 3 are build flags now.  This builds the library once per flag beside the
# product (RGC_LIB_OUT) and runs the short knob sequence of tests/test_gpu_parity.py on each build through RGC_HIP_LIB: every route
# must give the default build's poses.  GPU box:  bash scripts/exp_build_flags.sh        (builds there), or in two halves -- the builds
# where there is no GPU (hipcc cross-compiles), the runs on the GPU box:  bash scripts/exp_build_flags.sh build ; ... exp_build_flags.sh run
# (libraries in $RGC_FLAG_LIB_DIR, default exp_flags/ in the tree: git-ignored, shipped with the snapshot)
set -e
cd "$(dirname "$0")/.."
MODE=${1:-both}
D=${RGC_FLAG_LIB_DIR:-$PWD/exp_flags}
mkdir -p gpurun_out "$D"
FLAGS="-DRGC_SMALL_COPY=1 -DRGC_SRC_RES=0.5 -DRGC_SRC_RES=2.0 -DRGC_MAP_WIDE=0 -DRGC_MAP_WIDE=1000 -DRGC_MAP_WIDE_R=0 -DRGC_LM_POST=0 -DRGC_SOLVE_BEHIND_MAP=0 -DRGC_FE_SPEC=0 -DRGC_LM_SPARE_ASIDE=0"
if [ $MODE != run ]; then
  for flag in $FLAGS; do
    out=$D/librgc_alt_$(echo "$flag" | tr -c 'A-Za-z0-9' '_').so
    RGC_EXTRA_FLAGS="$flag" RGC_LIB_OUT="$out" python rgc-slam_amd/build.py > /dev/null
    rm -rf "${out%.so}_obj"
  done
fi
[ $MODE = build ] && exit 0
python - <<'PY' > gpurun_out/ref_poses.json
import json, sys, numpy as np
sys.path.insert(0, "scripts")
from knob_sequence import run
print(json.dumps([[T.tolist(), it, fit] for T, it, fit in run()]))
PY
for flag in $FLAGS; do
  out=$D/librgc_alt_$(echo "$flag" | tr -c 'A-Za-z0-9' '_').so
  RGC_HIP_LIB="$out" python - "$flag" <<'PY'
import json, sys, numpy as np
sys.path.insert(0, "scripts")
from knob_sequence import run
ref = json.load(open("gpurun_out/ref_poses.json"))
exact = not sys.argv[1].startswith("-DRGC_MAP_WIDE")
worst = 0.0
for (Ta, ia, fa), (Tb, ib, fb) in zip(ref, run()):
    d = float(np.abs(np.asarray(Ta, np.float32) - Tb).max())
    worst = max(worst, d)
    assert (d == 0.0 and ia == ib and fa == fb) if exact else (d <= 1e-6 and abs(fa - fb) <= 1e-6 * fa), (sys.argv[1], d, ia, ib, fa, fb)
print(sys.argv[1], "same poses" if exact else "poses within 1e-6", "max |dT| =", worst)
PY
done
