#!/bin/bash
# round 5, job 17: relu-bit words through a FIFO of whole layers: correctness, stamps (with and without ga0_chain), A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job17; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
for v in stbase stga2; do BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/libbhnerf_hip_$v.so python3 tools/dbg_chain_steps.py > $O/steps_$v.txt 2>&1; echo "== $v"; grep "take" $O/steps_$v.txt; done
bash tools/ab.sh libbhnerf_hip_base.so libbhnerf_hip.so 2>&1 | tee $O/ab.txt
