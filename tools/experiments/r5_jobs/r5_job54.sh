#!/bin/bash
# round 5, job 54: is the fused backward's +4 % behind the 12-wave forward the tape's group count?  W8 = this source with BHN_W12=0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job54b; mkdir -p $O
cd $R
for rep in 1 2; do for l in libbhnerf_hip_W8.so libbhnerf_hip_W8P.so libbhnerf_hip.so libbhnerf_hip_W12P.so; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done; done | tee $O/ab128.txt
