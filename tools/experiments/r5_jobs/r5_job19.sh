#!/bin/bash
# round 5, job 19: output layer on the vector ALU (OutDot): full GPU suite + A/B at 4x256 and 4x128
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job19; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
bash tools/ab.sh libbhnerf_hip_fifo.so libbhnerf_hip.so 2>&1 | tee $O/ab.txt
for l in libbhnerf_hip_fifo.so libbhnerf_hip.so libbhnerf_hip_fifo.so libbhnerf_hip.so; do echo -n "w128 $l "; BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/$l python bench.py --width 128 --steps 30 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(round(d['ms_per_step'],3), [round(v,3) for v in k.values()])"; done | tee -a $O/ab.txt
