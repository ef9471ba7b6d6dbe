#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job10; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
for v in stga2; do BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/libbhnerf_hip_$v.so python3 tools/dbg_chain_steps.py > $O/steps_$v.txt 2>&1; echo "== $v"; grep -A24 "^delta chain :" $O/steps_$v.txt | head -25; done
bash tools/ab.sh libbhnerf_hip_base.so libbhnerf_hip.so 2>&1 | tee $O/ab.txt
