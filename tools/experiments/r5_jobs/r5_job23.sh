#!/bin/bash
# round 5, job 23: resident weight image filled by LDS-DMA; identity arithmetic out of TrainStep.__call__: tests + config-5 times
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job23; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_api.py tests/test_gpu_fullsize_stokes.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
for B in 8 6 1; do python3 tools/cfg5_steps.py $B 300 | tail -1; python3 tools/cfg5_steps.py $B 300 graph | tail -1; done | tee $O/cfg5.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o k -- python3 $R/tools/cfg5_steps.py 8 200 > $O/run_b8.txt 2>&1
f=$(find /tmp/k5 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:6]: print('%-70s calls %5s  avg %8.1f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
