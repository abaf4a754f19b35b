#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/exp_dependent.py 40 > gpurun_out/s4_base_exp.txt 2>&1
bash scripts/prof_dependent.sh 30 0 > gpurun_out/s4_base_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s4_base_timeline.txt 2>&1
RGC_EXTRA_FLAGS="-DRGC_EXP_NO_TGT_EVENT" RGC_LIB_OUT=/tmp/librgc_a.so python rgc-slam_amd/build.py > /dev/null 2>&1
export RGC_HIP_LIB=/tmp/librgc_a.so
python scripts/exp_dependent.py 40 > gpurun_out/s4_noev_exp.txt 2>&1
bash scripts/prof_dependent.sh 30 0 > gpurun_out/s4_noev_kernels.txt 2>&1
python3 scripts/timeline_window.py gpurun_out/prof_dep 1100 3000 > gpurun_out/s4_noev_timeline.txt 2>&1
rm -rf gpurun_out/prof_dep
cat gpurun_out/s4_base_exp.txt gpurun_out/s4_noev_exp.txt
