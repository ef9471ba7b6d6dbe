#!/bin/bash
# round 4, GPU job 2: fused width-128 backward against the generic path (same inputs), tests, A/B timing
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
export TMPDIR=/tmp
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_nof128.so timeout 300 python3 tools/dbg_grad_ref.py save > $O/save.log 2>&1
cp gpurun_out/grad_ref.npz tools/_grad_ref.npz
timeout 300 python3 tools/dbg_grad_ref.py check > $O/check.log 2>&1; tail -30 $O/check.log
for r in 1 2; do for l in libbhnerf_hip_nof128.so libbhnerf_hip.so; do
  echo -n "$l " ; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l timeout 300 python3 bench.py --width 128 --steps 50 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(round(d['ms_per_step'],3), {n[:24]: round(v,3) for n,v in k.items()})"
done; done > $O/ab_w128.txt 2>&1
cat $O/ab_w128.txt
timeout 900 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize_stokes.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
