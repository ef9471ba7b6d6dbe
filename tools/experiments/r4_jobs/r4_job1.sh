#!/bin/bash
# round 4, GPU job 1: energy-budget rows under board telemetry, DROP_GA0 A/B (interleaved), 4x128 baseline
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
export TMPDIR=/tmp
hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/energy_bench.hip -o /tmp/energy_bench > $O/build.log 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibhnerf_amd/csrc -Iinclude tools/step_bench.hip -o /tmp/step_bench >> $O/build.log 2>&1
python3 tools/smi_sample.py > $O/telemetry.txt 2> /dev/null & SMI=$!
sleep 3
/tmp/energy_bench 5 > $O/rows_micro.txt 2>&1
/tmp/step_bench ceiling 5 > $O/rows_ring.txt 2>&1
python3 tools/energy_rows.py run 5 256 > $O/rows_lib256.txt 2> $O/rows_lib256.err
BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/libbhnerf_hip_dbg.so python3 tools/energy_rows.py run 5 256 > $O/rows_dbg256.txt 2> $O/rows_dbg256.err
python3 tools/energy_rows.py run 4 128 > $O/rows_lib128.txt 2> $O/rows_lib128.err
sleep 2; kill $SMI || true
python3 tools/energy_rows.py join $O/telemetry.txt $O/rows_micro.txt $O/rows_ring.txt $O/rows_lib256.txt $O/rows_dbg256.txt $O/rows_lib128.txt > $O/energy_table.txt 2>&1
cat $O/energy_table.txt
# DROP_GA0 A/B on this box, interleaved (3 rounds)
for r in 1 2 3; do for l in libbhnerf_hip.so libbhnerf_hip_ga0w20.so libbhnerf_hip_ga0w24.so libbhnerf_hip_ga0w30.so; do
  echo -n "$l " ; BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(round(d['ms_per_step'],3), {n[:24]: round(v,3) for n,v in k.items()})"
done; done > $O/ab_drop_ga0.txt 2>&1
cat $O/ab_drop_ga0.txt
python3 bench.py --width 128 --steps 50 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs > $O/bench_w128.json 2> $O/bench_w128.err
python3 -c "import json; d=json.load(open('$O/bench_w128.json')); print('w128', d['ms_per_step'], d['roofline']['kernel_ms'])"
