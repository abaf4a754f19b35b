#!/bin/bash
# kernel timeline of one f1 frame (rgc_mapreg_set_maps + rgc_mapreg_optimize)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mr_prof
rocprofv3 --kernel-trace -d /tmp/mr_prof -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_mapreg.py 6 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/timeline_any.py /tmp/mr_prof k_mapreg_associate | tail -70
