#!/bin/bash
# kernel times of the bf16 and 8-bit-tape backward at config 2's shape (rocprofv3 kernel trace of tools/dbg_t8.py time)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4t8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt8; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -o kt -- python3 $R/tools/dbg_t8.py time > $O/time.log 2>&1
f=$(find /tmp/kt8 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
cat $O/time.log | tail -4
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('chain_kernel', 'dw_kernel', 'reduce_kernel', 't8_')):
        print('%-90s calls %4s avg %9.1f us' % (n[:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
