#!/bin/bash
# round 5, job 40: drop_hd on the ring kernels with the emission's stores kept (4 bytes each): tests, A/B against HEAD (A)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job40; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab.sh libbhnerf_hip_A.so libbhnerf_hip.so 2>&1 | grep -v amdgpu | tee $O/ab.txt
