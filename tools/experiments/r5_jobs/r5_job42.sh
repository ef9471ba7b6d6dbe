#!/bin/bash
# round 5, job 42: randomised parity sweeps on the FINAL library (fused128 without h_depth on the tape)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job42; mkdir -p $O
cd $R
timeout 1500 python3 tools/fuzz_parity.py 250 605 2>&1 | grep -v amdgpu | grep -v "^general path" | tail -25 > $O/fuzz_fused.txt; tail -2 $O/fuzz_fused.txt
FUZZ_GENERAL=1 timeout 1500 python3 tools/fuzz_parity.py 60 606 2>&1 | grep -v amdgpu | grep -v "^general path" | grep -v "relu ties adj" | tail -5 > $O/fuzz_general.txt; tail -1 $O/fuzz_general.txt
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_nof128.so timeout 900 python3 tools/fuzz_fused128.py save 100 607 2>&1 | grep -v amdgpu | tail -1
timeout 900 python3 tools/fuzz_fused128.py check 100 607 2>&1 | grep -v amdgpu | tail -3 | tee $O/fuzz_fused128.txt
