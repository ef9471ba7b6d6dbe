#!/bin/bash
# round 5, job 33: general path variants (G) against the product build + parity of G
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job33; mkdir -p $O
cd $R
for l in libbhnerf_hip.so libbhnerf_hip_G.so; do echo $l; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l timeout 600 python3 tools/general_path_bench.py 8 3 2>&1 | grep general; done | tee $O/ab2.txt
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_G.so timeout 900 python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "outside_the_fused or general_path" 2>&1 | tail -2
