#!/bin/bash
# round 5, job 7: ga0_chain ablations (measurement builds): no consumer / no consumer + staging + enc DMA
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job7; mkdir -p $O
bash tools/ab.sh libbhnerf_hip_base.so libbhnerf_hip.so libbhnerf_hip_abl1.so libbhnerf_hip_abl7.so libbhnerf_hip_noga0.so 2>&1 | tee $O/ab_ga0c_abl.txt
