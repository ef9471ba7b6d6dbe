#!/bin/bash
# round 5, job 41: generic-path switches off (defaults), fused128 drop_hd on: full GPU suite + A/B both widths against HEAD (A)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job41; mkdir -p $O
cd $R
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab.sh libbhnerf_hip_A.so libbhnerf_hip.so 2>&1 | grep -v amdgpu | tee $O/ab.txt
for rep in 1 2; do for l in libbhnerf_hip_A.so libbhnerf_hip.so; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done; done | tee $O/ab128.txt
