#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k "hip_graph" > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 900 python3 - <<'PY' 2>&1 | tail -20
import json, sys, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.strong_scaling_share(torch.device('cuda:0')), indent=1))
PY
