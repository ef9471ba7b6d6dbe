#!/bin/bash
# round 5, job 48: long randomised parity sweep on the final library (md5 c69a69b4)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job48; mkdir -p $O
cd $R
timeout 3000 python3 tools/fuzz_parity.py 800 811 2>&1 | grep -v amdgpu | grep -v "^general path" | grep -v "relu ties adj" | tail -30 > $O/fuzz_fused.txt; tail -3 $O/fuzz_fused.txt
FUZZ_GENERAL=1 timeout 2400 python3 tools/fuzz_parity.py 150 812 2>&1 | grep -v amdgpu | grep -v "^general path" | grep -v "relu ties adj" | tail -5 > $O/fuzz_general.txt; tail -1 $O/fuzz_general.txt
