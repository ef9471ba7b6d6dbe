#!/bin/bash
# round 5, job 27: config-5 geometry at 64 frames/step: steady-state per-tile cost of the Stokes / compacted epilogue
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job27; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 64 8; do
rm -rf /tmp/k5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o k -- python3 $R/tools/cfg5_steps.py $B 60 > $O/run_b$B.txt 2>&1
f=$(find /tmp/k5 -name "*kernel_stats.csv" | head -1); echo "B=$B"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:5]: print('%-70s calls %5s  avg %8.1f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
done 2>&1 | tee $O/stats.txt
