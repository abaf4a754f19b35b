#!/bin/bash
# A/B of environment knobs on the bench's c3 / c5 configurations: bash scripts/ab_env_cfg.sh <rounds> "A=1" "-" ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for v in "$@"; do
if [ "$v" = "-" ]; then e="A=1"; else e="$v"; fi
env $e timeout 900 python bench.py --steps 10 --warmup 2 --configs c3,c5 --no-cpu-baseline --no-two-sequences 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]', 'value', d['value'], [(c['config'][:2], c['scans_per_s'], c['one_frame_at_a_time_scans_per_s'], c.get('lazy_target_scans_per_s')) for c in d['configs']])"
done
done
