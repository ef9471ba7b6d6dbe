#!/bin/bash
# round 5, job 37: LBITS job-weight sweep, higher weights; the h-tile form (debug build of HEAD: libbhnerf_hip_dbgA.so) on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job37; mkdir -p $O
cd $R
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_dbgA.so SWEEP=jobs JL="7" J1="13" timeout 600 python3 tools/dbg_dw_grid.py 2>&1 | grep -v amdgpu.ids | grep -E "JOBL|dw" | paste - - | tee $O/sweep2.txt
SWEEP=jobs JL="6 8 10 12 14 16" J1="12 13 14" timeout 1500 python3 tools/dbg_dw_grid.py 2>&1 | grep -v amdgpu.ids | grep -E "JOBL|dw" | paste - - | tee -a $O/sweep2.txt
