#!/bin/bash
# round 5, job 21: forward rings copy the hidden fragments only (resident encoded-input block): tests + A/B (4x256, 8x256 inference, 4x128 unaffected)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job21; mkdir -p $O
timeout 3000 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_stokes.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab.sh libbhnerf_hip_head.so libbhnerf_hip.so 2>&1 | tee $O/ab.txt
L=$PWD/bhnerf_amd/csrc
for r in 1 2; do for l in libbhnerf_hip_head.so libbhnerf_hip.so; do BHNERF_HIP_LIB=$L/$l python3 tools/ab_infer.py 256 4 2>&1 | tail -1; BHNERF_HIP_LIB=$L/$l python3 tools/ab_infer.py 256 8 2>&1 | tail -1; done; done | tee -a $O/ab.txt
