#!/bin/bash
# round 5, job 47: chi2 of a handful of small planes in one block (no loss_sum launch): tests + config-5 one-frame share
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job47; mkdir -p $O
cd $R
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -2 $O/tests.log
for rep in 1 2; do python3 tools/cfg5_steps.py 1 400 graph | tail -1; done 2>&1 | grep -v amdgpu | tee $O/cfg5.txt
