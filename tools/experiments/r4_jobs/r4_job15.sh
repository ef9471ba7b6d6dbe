#!/bin/bash
# board power / clock per kernel group, bf16 against the 8-bit tape mode (tools/t8_telemetry.py under tools/smi_sample.py)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4t8; mkdir -p $O; cd $R
python3 tools/smi_sample.py > $O/tele.txt & SMI=$!
sleep 2; timeout 200 python3 tools/t8_telemetry.py 4 > $O/rows.txt 2>&1; sleep 1; kill $SMI
python3 - $O/tele.txt $O/rows.txt <<'PY' | tee $O/t8_telemetry.txt
import sys
tele = [l.split() for l in open(sys.argv[1]) if l[0] != '#']
tele = [(float(a[0]), float(a[1]), float(a[2])) for a in tele]
print('# idle: %.0f W' % min(p for _, p, _ in tele))
print('%-8s %-10s %9s %8s %9s %12s' % ('mode', 'row', 'ms/call', 'board W', 'SMI MHz', 'J per call'))
for l in open(sys.argv[2]):
    if not l.startswith('ROW'):
        continue
    _, mode, name, t0, t1, n, ms = l.split()
    w = [(p, c) for t, p, c in tele if float(t0) + 0.7 < t < float(t1) - 0.3]
    if not w:
        continue
    pw = sum(p for p, _ in w) / len(w); ck = sum(c for _, c in w) / len(w)
    print('%-8s %-10s %9.3f %8.0f %9.0f %12.2f' % (mode, name, float(ms), pw, ck, pw * float(ms) * 1e-3))
PY
