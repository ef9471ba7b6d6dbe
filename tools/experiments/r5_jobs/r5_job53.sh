#!/bin/bash
# round 5, job 53: 12-wave forward pair of the fused 4x128 path for ray sets of >= 3072 groups per frame: suite, A/B, config 5
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job53; mkdir -p $O
cd $R
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -2 $O/tests.log
for rep in 1 2; do for l in libbhnerf_hip_A.so libbhnerf_hip.so; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done; done | tee $O/ab128.txt
for l in libbhnerf_hip_A.so libbhnerf_hip.so; do for B in 8 1; do echo -n "$l "; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 tools/cfg5_steps.py $B 300 graph | tail -1; done; done 2>&1 | grep -v amdgpu | tee $O/cfg5_ab.txt
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_A.so python3 tools/ab_infer.py 128 4 2>&1 | grep -v amdgpu | tail -1 | tee $O/infer.txt; python3 tools/ab_infer.py 128 4 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/infer.txt
