#!/bin/bash
# round 5, job 31: randomised parity sweeps on the final library: fused shapes (new seed), general shapes, the fused 4x128 backward
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job31; mkdir -p $O
cd $R
timeout 1500 python3 tools/fuzz_parity.py 250 505 2>&1 | grep -v amdgpu | grep -v "^general path" | tail -25 > $O/fuzz_fused.txt; tail -3 $O/fuzz_fused.txt
FUZZ_GENERAL=1 timeout 1500 python3 tools/fuzz_parity.py 80 506 2>&1 | grep -v amdgpu | grep -v "^general path" | tail -25 > $O/fuzz_general.txt; tail -3 $O/fuzz_general.txt
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_nof128.so timeout 900 python3 tools/fuzz_fused128.py save 60 507 2>&1 | grep -v amdgpu | tail -2
timeout 900 python3 tools/fuzz_fused128.py check 60 507 2>&1 | grep -v amdgpu | tail -5 | tee $O/fuzz_fused128.txt
