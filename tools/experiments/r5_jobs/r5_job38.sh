#!/bin/bash
# round 5, job 38: h_depth off the tape at width 256 too (forward stores nothing for it; dW job from the relu bits): tests, A/B, sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job38; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_api.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_stokes.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
bash tools/ab.sh 2>&1 | grep -v amdgpu | tee $O/ab.txt
SWEEP=jobs JL="4 6 8 10" J1="12 13 14" timeout 1500 python3 tools/dbg_dw_grid.py 2>&1 | grep -v amdgpu.ids | grep -E "JOBL|dw" | paste - - | tee $O/sweep.txt
