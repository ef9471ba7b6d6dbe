#!/bin/bash
# round 5, job 1: half-size workgroups (two per CU) in the 4x256 inference forward: correctness + A/B on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job1; mkdir -p $O
L=$PWD/bhnerf_amd/csrc
BHNERF_HIP_LIB=$L/libbhnerf_hip_half.so timeout 900 python3 -m pytest tests/test_gpu_forward.py -x -q -m gpu > $O/test_half.log 2>&1; echo "tests rc $?" >> $O/test_half.log
for r in 1 2 3; do
  for l in libbhnerf_hip.so libbhnerf_hip_half.so; do BHNERF_HIP_LIB=$L/$l python3 tools/ab_infer.py 256 4 2>&1 | tail -1; done
done > $O/ab_infer.txt
for l in libbhnerf_hip.so libbhnerf_hip_half.so; do BHNERF_HIP_LIB=$L/$l python3 tools/ab_infer.py 256 8 2>&1 | tail -1; done >> $O/ab_infer.txt
tail -3 $O/test_half.log; cat $O/ab_infer.txt
