#!/bin/bash
# A/B of prebuilt libraries on the bench's c3 / c5 configurations and the headline: bash scripts/ab_lib_cfg.sh <rounds> cur head ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for name in "$@"; do
if [ $name = cur ]; then unset RGC_HIP_LIB; else export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_flags/librgc_$name.so; fi
timeout 900 python bench.py --steps 20 --warmup 4 --configs c3,c5 --no-cpu-baseline --no-two-sequences 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'value', d['value'], 'steady', d['steady_state']['two_contexts']['ms_per_step'], [(c['config'][:2], c['scans_per_s'], c['one_frame_at_a_time_scans_per_s'], c.get('lazy_target_scans_per_s')) for c in d['configs']])"
done
done
