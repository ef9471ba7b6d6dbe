#!/bin/bash
# round 5, job 6: dW_0 accumulated by the delta chain (ga0_chain): correctness + A/B on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job6; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -6 $O/tests.log
bash tools/ab.sh libbhnerf_hip_base.so libbhnerf_hip.so 2>&1 | tee $O/ab_ga0c.txt
