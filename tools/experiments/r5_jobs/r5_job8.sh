#!/bin/bash
# round 5, job 8: ring-step stamps of the delta chain with and without ga0_chain (stamps builds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job8; mkdir -p $O
for v in stbase stga; do BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/libbhnerf_hip_$v.so python3 tools/dbg_chain_steps.py > $O/steps_$v.txt 2>&1; done
grep -A26 "^delta chain :" $O/steps_stbase.txt | head -28; grep -A26 "^delta chain :" $O/steps_stga.txt | head -28
