#!/bin/bash
# round 5, job 39: A = before (h_depth tiles on the tape), L = dW job from the relu bits only, product = + forward stores no h_depth
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job39; mkdir -p $O
cd $R
bash tools/ab.sh libbhnerf_hip_A.so libbhnerf_hip_L.so libbhnerf_hip.so 2>&1 | grep -v amdgpu | tee $O/ab.txt
