#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k "hip_graph" 2>&1 | tail -3
timeout 600 python3 - <<'PY' 2>&1 | grep "wall_\|gpu_ms\|rror"
import json, sys, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.strong_scaling_share(torch.device('cuda:0')), indent=1))
PY
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_nof128.so timeout 1500 python3 tools/fuzz_fused128.py save 300 7 2>&1 | tail -2
timeout 1500 python3 tools/fuzz_fused128.py check 300 7 > $O/fuzz_fused128.txt 2>&1; tail -5 $O/fuzz_fused128.txt; rm -f gpurun_out/fuzz_fused128_ref.npz
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
