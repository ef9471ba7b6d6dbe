#!/bin/bash
# round 5, job 46: fused-128 training forward without the relu-bit extraction of the tiles whose bits nobody records: tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job46; mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_api.py tests/test_gpu_fullsize_stokes.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -2 $O/tests.log
for rep in 1 2 3; do for l in libbhnerf_hip_A.so libbhnerf_hip.so; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done; done | tee $O/ab128.txt
bash tools/ab.sh libbhnerf_hip_A.so libbhnerf_hip.so 2>&1 | grep -v amdgpu | tee $O/ab256.txt
