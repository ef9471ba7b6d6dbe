#!/bin/bash
# round 5, job 11: full GPU test suite + smoke + bench line with the delta chain accumulating dW_0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job11; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -5 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -6 $O/smoke.log
python3 bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['step_mfma_frac']); print('w128', d['width128']['ms_per_step'], d['width128'].get('ms_per_step_hip_graph')); print({k:(v.get('ms_per_step'), v.get('ms_per_step_hip_graph')) for k,v in d['other_configs'].items()}); print('t8', d['tape8_mode']['ms_per_step'], 'f32', d['parity_mode']['ms_per_step'])"
