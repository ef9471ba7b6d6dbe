#!/bin/bash
# round 5, job 22: per-kernel times of config 5's step (8 frames and 1 frame)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 8 1; do
rm -rf /tmp/k5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o k -- python3 $R/tools/cfg5_steps.py $B 200 > $O/run_b$B.txt 2>&1
f=$(find /tmp/k5 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_b$B.csv
tail -1 $O/run_b$B.txt; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:16]: print('%-70s calls %5s  avg %8.1f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
done
