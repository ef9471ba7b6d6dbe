#!/bin/bash
# round 5, job 5: the delta chain's ring copies 16 instead of 18 fragments per chunk -- correctness + A/B on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job5; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
bash tools/ab.sh libbhnerf_hip_base.so libbhnerf_hip.so 2>&1 | tee $O/ab_chain16.txt
