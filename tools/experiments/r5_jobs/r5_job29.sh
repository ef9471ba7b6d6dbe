#!/bin/bash
# round 5, job 29: bhn_geom.ray_span (per-wave ray sums on compacted layouts): tests + config-5 times + config-3 step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job29; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
for B in 8 64 1; do python3 tools/cfg5_steps.py $B 200 | tail -1; done 2>&1 | grep -v amdgpu | tee $O/cfg5.txt
python3 tools/cfg5_steps.py 1 300 graph | tail -1 | tee -a $O/cfg5.txt
