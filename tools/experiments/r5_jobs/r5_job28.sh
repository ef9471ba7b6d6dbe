#!/bin/bash
# round 5, job 28: Stokes weights loaded in front of the ray-segment loop: A (before) / B (after): config 5 at 8 / 64 frames, the
# 4x256 headline, forward tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job28b; mkdir -p $O
cd $R
true
for rep in 1 2; do for l in libbhnerf_hip_A.so libbhnerf_hip.so; do for B in 8 64; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 tools/cfg5_steps.py $B 200 | tail -1; done; done; done 2>&1 | grep -v amdgpu | tee $O/cfg5_ab.txt

