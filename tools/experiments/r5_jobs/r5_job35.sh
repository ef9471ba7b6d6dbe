#!/bin/bash
# round 5, job 35: HIP-graph step on the general path + full GPU suite on the current library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job35; mkdir -p $O
cd $R
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -5 $O/tests.log
