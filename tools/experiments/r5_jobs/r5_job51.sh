#!/bin/bash
# round 5, job 51: distance of the h_2 / h_3 tape tensors of the fused 4x128 path: multiple of 64 MiB (product) against 4 KiB (T); A = before the 12-wave forward
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job51; mkdir -p $O
cd $R
for rep in 1 2 3; do for l in libbhnerf_hip_A.so libbhnerf_hip_T.so libbhnerf_hip.so; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done; done | tee $O/ab128.txt
