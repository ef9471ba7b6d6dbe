#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job18; mkdir -p $O
bash tools/ab.sh libbhnerf_hip_base.so libbhnerf_hip_fifo.so libbhnerf_hip.so 2>&1 | tee $O/ab.txt
