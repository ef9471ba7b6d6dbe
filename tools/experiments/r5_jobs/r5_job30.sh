#!/bin/bash
# round 5, job 30: which kernel of the general path dominates (4x128 deg 5, 4x512 deg 3)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job30; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kg -o k -- python3 $R/tools/general_path_bench.py 8 2 > $O/run.txt 2>&1
f=$(find /tmp/kg -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY' | tee $O/stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:12]: print('%-80s calls %5s  avg %10.1f us  %5s %%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
cat $O/run.txt | grep -v amdgpu
