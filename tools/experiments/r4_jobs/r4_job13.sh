#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python3 tools/dbg_stress.py 2>&1 | tail -4
