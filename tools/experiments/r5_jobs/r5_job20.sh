#!/bin/bash
# round 5, job 20: final library: full GPU suite + smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job20; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -6 $O/smoke.log
