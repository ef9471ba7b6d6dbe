#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_st128.so timeout 300 python3 tools/dbg_bwd128_stamps.py > $O/stamps.txt 2>&1
head -31 $O/stamps.txt
bash tools/r4_job5.sh 2>&1 | grep "lib\|w128d4\|OK\|FAIL"
