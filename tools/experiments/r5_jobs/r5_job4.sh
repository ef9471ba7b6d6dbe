#!/bin/bash
# round 5, job 4: re-run of the new parity tests + upper bound of "dW_0 accumulated by the delta chain" (no gA_0 on the tape)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job4; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_fullsize_stokes.py tests/test_gpu_backward.py -q -m gpu -k "oracle_on_a_ray_subset or batch_of_six" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
grep -E "^\[|passed|failed|rc" $O/tests.log | tail -20
bash tools/ab.sh libbhnerf_hip.so libbhnerf_hip_noga0.so 2>&1 | tee $O/ab_noga0.txt
