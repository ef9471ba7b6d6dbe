#!/bin/bash
# round 5, job 3: the new parity tests (oracle gradients at config 3 / 5 shape, 4x128 without skip / batch of 6, 8-bit tape vs oracle)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job3; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_fullsize_stokes.py tests/test_gpu_fullsize.py tests/test_gpu_backward.py -q -m gpu -k "oracle_on_a_ray_subset or batch_of_six" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
grep -E "^\[|passed|failed|rc" $O/tests.log | tail -40
