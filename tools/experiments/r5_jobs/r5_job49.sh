#!/bin/bash
# round 5, job 49: width-128 inference forward on 12-wave workgroups (variant X) against the product build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job49; mkdir -p $O
cd $R
for rep in 1 2 3; do for l in libbhnerf_hip.so libbhnerf_hip_X.so; do BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 tools/ab_infer.py 128 4 2>&1 | grep -v amdgpu | tail -1; done; done | tee $O/ab.txt
