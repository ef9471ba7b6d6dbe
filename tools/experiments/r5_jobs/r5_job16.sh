#!/bin/bash
# round 5, job 16: delta-chain ring-step stamps under the three cache policies of the tape stores (debug build: BHN_DEBUG_POLICY 0 nt / 1 plain / 2 sc1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job16; mkdir -p $O
for pol in 0 1 2; do BHN_DEBUG_POLICY=$pol BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/libbhnerf_hip_stga2.so python3 tools/dbg_chain_steps.py > $O/steps_pol$pol.txt 2>&1; echo "== policy $pol"; grep "take" $O/steps_pol$pol.txt; done
