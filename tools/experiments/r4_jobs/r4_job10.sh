#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
timeout 900 python3 tools/dbg_stress.py > $O/stress.txt 2>&1; tail -6 $O/stress.txt
timeout 3000 python3 tools/fuzz_parity.py 1000 4 > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt; grep -c "adjudicated" $O/fuzz.txt
