#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4g/bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
r = d['roofline']; print({k: r[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'step_mfma_frac')}); print(r['kernels'])
w = d.get('width128'); print('width128', {k: w[k] for k in w if k != 'roofline'} if w else None)
if w and 'roofline' in w: print(w['roofline']['kernels'], w['roofline']['step_mfma_frac'])
print('cpu', d.get('cpu_baseline'))
print('other', d.get('other_configs'))
PY
