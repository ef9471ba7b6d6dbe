#!/bin/bash
# round 5, job 32: general-shape fuzz again (f32 / bf16 mode agreement to rounding instead of bitwise)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job32; mkdir -p $O
cd $R
FUZZ_GENERAL=1 timeout 1500 python3 tools/fuzz_parity.py 80 506 2>&1 | grep -v amdgpu | grep -v "^general path" | grep -v "relu ties adj" | tail -25 > $O/fuzz_general.txt; tail -12 $O/fuzz_general.txt | cut -c1-300
