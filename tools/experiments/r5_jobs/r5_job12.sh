#!/bin/bash
# round 5, job 12: dW job weights re-tuned without the layer-0 job (debug build: BHN_DEBUG_JOB1_W / BHN_DEBUG_JOBL_W)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job12; mkdir -p $O
SWEEP=jobs JL="4 6 8 10 12" J1="9 11 13 15" python3 tools/dbg_dw_grid.py 2>&1 | grep -E "JOBL_W|dw " | paste - - > $O/dw_job_weights.txt; cat $O/dw_job_weights.txt
