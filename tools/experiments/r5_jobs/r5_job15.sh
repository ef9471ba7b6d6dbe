#!/bin/bash
# round 5, job 15: the tail of tools/collect_profiles.sh (ring ceiling microbench + telemetry) after step_bench.hip was repaired
set -uo pipefail
TAG=r5
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$R/bhnerf_amd/csrc -I$R/include $R/tools/step_bench.hip -o /tmp/step_bench > /tmp/step_bench.log 2>&1
: > $O/${TAG}_telemetry_ceiling.txt
python3 $R/tools/smi_sample.py >> $O/${TAG}_telemetry_ceiling.txt & SMI=$!
sleep 2; /tmp/step_bench ceiling 8 > $O/${TAG}_ring_ceiling_microbench.txt 2>&1; sleep 1; kill $SMI || true
: > $O/${TAG}_telemetry_bench.txt
python3 $R/tools/smi_sample.py >> $O/${TAG}_telemetry_bench.txt & SMI=$!
sleep 2; date +"# bench.py --steps 400 starts %s" >> $O/${TAG}_telemetry_bench.txt
python3 $R/bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/bench400.json 2> /tmp/bench400.err
date +"# bench.py ends %s" >> $O/${TAG}_telemetry_bench.txt; sleep 1; kill $SMI || true
python3 -c "import json; d=json.load(open('/tmp/bench400.json')); print('# bench.py --steps 400: ms_per_step %.3f' % d['ms_per_step']); print('# roofline:', json.dumps({k: d['roofline'].get(k) for k in ('kernel','frac','traffic','profiles_match_this_build','step_mfma_frac','step_mfma_busy_frac_from_profiles','step_mlp_traffic_from_profiles','lds_bank_conflict_frac_from_profiles')}))" >> $O/${TAG}_telemetry_bench.txt
tail -3 $O/${TAG}_telemetry_bench.txt; cat $O/${TAG}_ring_ceiling_microbench.txt | tail -8
