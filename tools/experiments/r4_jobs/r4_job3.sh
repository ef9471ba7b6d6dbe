#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_st128.so timeout 300 python3 tools/dbg_bwd128_stamps.py > $O/stamps.txt 2>&1
cat $O/stamps.txt | head -90
timeout 900 bash tools/pmc_collect.sh bf16 128 > $O/pmc.log 2>&1
cat $R/gpurun_out/pmc/pass*.txt > $O/sq128.txt; grep -A40 bwd128 $O/sq128.txt | head -120
