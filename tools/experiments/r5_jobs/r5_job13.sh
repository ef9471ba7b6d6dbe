#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job13; mkdir -p $O
python3 bench.py --steps 40 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['kernel_ms']); print('w128', d['width128']['ms_per_step'], d['width128'].get('ms_per_step_hip_graph')); print({k:(v.get('ms_per_step'), v.get('ms_per_step_hip_graph')) for k,v in d['other_configs'].items()}); print(d['strong_scaling_share'])"
timeout 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -3
