#!/bin/bash
# round 5, job 25: general-shape path timing + full GPU suite on the library with general_mlp.hip
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job25; mkdir -p $O
cd $R
timeout 600 python3 tools/general_path_bench.py 8 3 2>&1 | grep -v amdgpu.ids | tee $O/general_bench.txt
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
