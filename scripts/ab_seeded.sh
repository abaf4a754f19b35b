#!/bin/bash
# A/B of prebuilt libraries (exp_flags/librgc_<name>.so, "cur" = the product) on the seeded launch of the map's kNN kernel:
#   bash scripts/ab_seeded.sh <rounds> <name> ...      -> per library the launch's time (profiling region, frames 2..5 of a re-framed 1 M-point map)
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for name in "$@"; do
if [ $name = cur ]; then unset RGC_HIP_LIB; else export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_flags/librgc_$name.so; fi
timeout 200 python scripts/lab_seeded.py 1000000 6 2>/dev/null | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$name', 'first', rows[0]['ms']['knn_cov_target'], 'seeded', [r['ms']['knn_cov_target'] for r in rows[2:]])"
done
done
