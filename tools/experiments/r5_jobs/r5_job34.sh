#!/bin/bash
# round 5, job 34: per-kernel times of the general path, one shape at a time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job34; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 2 4; do
export GEN_ONLY=$c
rm -rf /tmp/kg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kg -o k -- python3 $R/tools/general_path_bench.py 8 2 > $O/run$c.txt 2>&1
grep general $O/run$c.txt
f=$(find /tmp/kg -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:5]: print('%-80s calls %5s  avg %10.1f us  total %8.1f ms %5s %%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
done 2>&1 | tee $O/stats.txt
