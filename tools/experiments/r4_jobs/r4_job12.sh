#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
echo "--- round-3 tree, draws 335 553"; (cd tools/_r3tree && FUZZ_ONLY=335,553 timeout 600 python3 tools/fuzz_parity.py 1000 4 2>&1 | tail -4)
echo "--- this tree, draws 335 553"; FUZZ_ONLY=335,553 timeout 600 python3 tools/fuzz_parity.py 1000 4 2>&1 | tail -4
bash tools/r4_job11.sh
