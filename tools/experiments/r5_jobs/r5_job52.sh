#!/bin/bash
# round 5, job 52: rocprofv3 kernel averages of the 4x128 step, A (8-wave forward pair) against the product (12-wave)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_job52; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for l in libbhnerf_hip_A.so libbhnerf_hip.so; do
export BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l
rm -rf /tmp/kw; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kw -o k -- python3 $R/bench.py --width 128 --steps 20 --warmup 3 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/kw.log 2>&1
f=$(find /tmp/kw -name "*kernel_stats.csv" | head -1); echo $l; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:7]: print('  %-72s calls %4s  avg %9.1f us' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done 2>&1 | tee $O/stats.txt
