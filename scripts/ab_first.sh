#!/bin/bash
# first (list-building) frame of the map's kNN launch per prebuilt library: bash scripts/ab_first.sh <rounds> <name> ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for name in "$@"; do
if [ $name = cur ]; then unset RGC_HIP_LIB; else export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_flags/librgc_$name.so; fi
timeout 200 python scripts/lab_seeded.py 1000000 4 2>/dev/null | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$name', 'searched', [r.get('searched') for r in rows[:3]], 'knn', [r['ms']['knn_cov_target'] for r in rows], [r['ms'] for r in rows[:2]])"
done
done
