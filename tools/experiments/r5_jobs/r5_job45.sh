#!/bin/bash
# round 5, job 45: output row summed by the last reduce128 block (no third kernel): tests, fused-vs-generic fuzz, config-5 share
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job45; mkdir -p $O
cd $R
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -2 $O/tests.log
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_nof128.so timeout 900 python3 tools/fuzz_fused128.py save 100 707 2>&1 | grep -v amdgpu | tail -1
timeout 900 python3 tools/fuzz_fused128.py check 100 707 2>&1 | grep -v amdgpu | tail -2 | tee $O/fuzz_fused128.txt
for B in 8 1; do python3 tools/cfg5_steps.py $B 300 | tail -1; python3 tools/cfg5_steps.py $B 300 graph | tail -1; done 2>&1 | grep -v amdgpu | tee $O/cfg5.txt
