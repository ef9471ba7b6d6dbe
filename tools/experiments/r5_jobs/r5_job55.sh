#!/bin/bash
# round 5, job 55: fused backward on tapes of 8- and 12-group tiles at a size both divide (132 x 132 rays: 34848 groups per frame) and at config 2's (128 x 128)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job55; mkdir -p $O
cd $R
for img in 132 128; do for rep in 1 2; do for l in libbhnerf_hip_W8.so libbhnerf_hip.so; do echo -n "image $img $l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --image $img --width 128 --steps 30 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done; done; done | tee $O/ab.txt
