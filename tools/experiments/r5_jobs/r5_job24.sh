#!/bin/bash
# round 5, job 24: general-shape path (posenc_deg > 4, net_width > 256): first GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job24; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu -s -k "outside_the_fused or general_path" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
grep -E "general path|passed|failed|Error|error|assert" $O/tests.log | head -60
