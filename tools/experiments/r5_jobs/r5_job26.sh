#!/bin/bash
# round 5, job 26: dW job weights re-swept on the ga0_chain structure (no layer-0 job)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job26; mkdir -p $O
cd $R
for rep in 1 2; do SWEEP=jobs JL="5 6 7 8" J1="11 12 13" timeout 1500 python3 tools/dbg_dw_grid.py 2>&1 | grep -v amdgpu.ids | grep -E "JOBL|dw" | paste - - | tee -a $O/sweep2.txt; done
