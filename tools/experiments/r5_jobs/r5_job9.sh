#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job9; mkdir -p $O
for v in stbase stbase4 stabl7 stabl1 stga; do BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/libbhnerf_hip_$v.so python3 tools/dbg_chain_steps.py > $O/steps_$v.txt 2>&1; echo "== $v"; grep -A8 "^delta chain :" $O/steps_$v.txt | head -9; done
