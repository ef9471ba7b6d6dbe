#!/bin/bash
# round 5, job 43: fused 4x128 backward against the generic one again (output row compared at the bf16-rounding level)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job43; mkdir -p $O
cd $R
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_nof128.so timeout 900 python3 tools/fuzz_fused128.py save 100 607 2>&1 | grep -v amdgpu | tail -1
timeout 900 python3 tools/fuzz_fused128.py check 100 607 2>&1 | grep -v amdgpu | tail -3 | tee $O/fuzz_fused128.txt
