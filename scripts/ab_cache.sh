#!/bin/bash
# A/B of the neighbour-list cache (RGC_KNN_CACHE=0/1 in the environment, one library) on the seeded launch of the map's kNN kernel:
#   bash scripts/ab_cache.sh <rounds>
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $1); do
for v in 0 1; do
RGC_KNN_CACHE=$v timeout 200 python scripts/lab_seeded.py 1000000 6 2>/dev/null | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('cache=$v', 'searched', [r.get('searched') for r in rows[:3]], 'first', rows[0]['ms']['knn_cov_target'], 'later', [r['ms']['knn_cov_target'] for r in rows[2:]], 'grid', [r['ms'].get('grid') for r in rows[1:4]])"
done
done
