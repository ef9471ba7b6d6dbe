#!/bin/bash
# Everything under profiles/ comes from this script (run on the GPU box: gpurun -- bash tools/collect_profiles.sh [tag]):
#   <tag>_bench_line.json         the bench line of `python3 bench.py`
#   <tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command (per-kernel averages)
#   <tag>_f32_kernel_stats.csv    the same for `bench.py --mode f32` (the parity mode)
#   <tag>_t8_kernel_stats.csv     the same for tools/debug/dbg_t8.py time: the bf16 and the 8-bit-tape kernels side by side
#   <tag>_pmc_traffic.json        HBM traffic per launch (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, gfx950 x2 on FETCH_SIZE)
#   <tag>_sq_counters.txt / <tag>_sq_summary.json   SQ counter passes of tools/pmc_run.py (MFMA busy, waits, LDS conflicts)
#   <tag>_ring_ceiling_microbench.txt + <tag>_telemetry_ceiling.txt   tools/step_bench.hip `ceiling`: the library's ring step
#                                 with and without the tape stores, in-kernel clock, board power / SMI clock sampled beside it
#   <tag>_telemetry_bench.txt     board power / SMI clock during 400 training steps of bench.py (tools/smi_sample.py)
# The files land in gpurun_out/profiles/ (merged back by gpurun); copy them to profiles/ and commit.
set -euo pipefail
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles
mkdir -p $O
need() { [ -s "$1" ] || { echo "collect_profiles: missing or empty $1" >&2; exit 1; }; }
# TAIL_ONLY=1: only the ring-ceiling micro-benchmark and the telemetry runs at the end (they do not depend on the passes above)
if [ -z "${TAIL_ONLY:-}" ]; then
python3 $R/bench.py > $O/${TAG}_bench_line.json 2> $O/bench.err
need $O/${TAG}_bench_line.json
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); need "$f"; cp "$f" $O/${TAG}_bench_kernel_stats.csv
rm -rf /tmp/ktf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktf -o kt -- python3 $R/bench.py --mode f32 --steps 3 --warmup 1 --no-cpu-baseline --no-tutorial-domain --no-other-configs --no-width128 > /tmp/ktf.log 2>&1
f=$(find /tmp/ktf -name "*kernel_stats.csv" | head -1); need "$f"; cp "$f" $O/${TAG}_f32_kernel_stats.csv
rm -rf /tmp/ktw; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktw -o kt -- python3 $R/bench.py --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/ktw.log 2>&1
f=$(find /tmp/ktw -name "*kernel_stats.csv" | head -1); need "$f"; cp "$f" $O/${TAG}_w128_kernel_stats.csv
# bf16 against the 8-bit tape mode (BHN_BF16_T8), kernel by kernel, at config 2's shape
rm -rf /tmp/kt8; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -o kt -- python3 $R/tools/debug/dbg_t8.py time > $O/${TAG}_t8_times.txt 2>&1
f=$(find /tmp/kt8 -name "*kernel_stats.csv" | head -1); need "$f"; cp "$f" $O/${TAG}_t8_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/pm_$c.log 2>&1
  f=$(find /tmp/pm_$c -name "*counter_collection.csv" | head -1); need "$f"; cp "$f" /tmp/pm_$c.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pw_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pw_$c -o p -- python3 $R/bench.py --width 128 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/pw_$c.log 2>&1
  f=$(find /tmp/pw_$c -name "*counter_collection.csv" | head -1); need "$f"; cp "$f" /tmp/pw_$c.csv
done
LIBMD5=$(md5sum $R/bhnerf_amd/csrc/libbhnerf_hip.so | cut -d" " -f1)
export LIBMD5
export REPO_ROOT=$R
for W in 256 128; do
export PMC_W=$W
python3 - > $O/${TAG}_pmc_traffic$( [ $W = 128 ] && echo _w128 ).json <<'PY'
import csv, json, collections, os
PFX = '/tmp/pm_' if os.environ.get('PMC_W') == '256' else '/tmp/pw_'
out = collections.OrderedDict()
ms = collections.defaultdict(list)
import sys
sys.path.insert(0, os.environ['REPO_ROOT'] + '/tools')
from kernel_names import short_name as short, expected_kernels
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(PFX + '%s.csv' % c)):
        n = short(r['Kernel_Name'])
        if n and r['Counter_Name'] == c:
            acc[n].append(float(r['Counter_Value']))
            ms[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    for n, v in acc.items():
        out.setdefault(n, {})[c + '_KiB'] = round(sum(v) / len(v), 1)
for n, d in out.items():
    d['hbm_bytes'] = int((2 * d.get('FETCH_SIZE_KiB', 0) + d.get('WRITE_SIZE_KiB', 0)) * 1024)
    d['ms_under_profiler'] = round(sum(ms[n]) / len(ms[n]), 4)
missing = [k for k in expected_kernels(int(os.environ['PMC_W'])) if k not in out or 'FETCH_SIZE_KiB' not in out[k] or 'WRITE_SIZE_KiB' not in out[k]]
if missing:          # (round 4: a renamed template argument dropped both chain kernels from this file without a word)
    sys.exit('collect_profiles: kernels missing from the traffic passes: %s (names seen: %s)' % (missing, sorted(out)))
step = [k for k in expected_kernels(int(os.environ['PMC_W'])) if 'inference' not in k]
out['step_mlp_kernels_sum'] = {'kernels': step, 'hbm_bytes': sum(out[k]['hbm_bytes'] for k in step)}
print(json.dumps({'lib_md5': os.environ.get('LIBMD5'), 'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128`; averages per launch in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 correction (FETCH_SIZE reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section)', 'kernels': out}, indent=1))
PY
done
need $O/${TAG}_pmc_traffic.json
need $O/${TAG}_pmc_traffic_w128.json
for W in 256 128; do
export PMC_W=$W
SFX=$( [ $W = 128 ] && echo _w128 || true )
bash $R/tools/pmc_collect.sh bf16 $W > /tmp/sq.log 2>&1
ls $R/gpurun_out/pmc/pass*.txt > /dev/null
cat $R/gpurun_out/pmc/pass*.txt > $O/${TAG}${SFX}_sq_counters.txt
python3 - $O/${TAG}${SFX}_sq_counters.txt > $O/${TAG}${SFX}_sq_summary.json <<'PY'
import sys, json, re, collections, os
k = None; d = collections.OrderedDict()
for l in open(sys.argv[1]):
    if not l.startswith(' '):
        k = l.strip(); d.setdefault(k, {})
    else:
        m = re.match(r'\s+(\S+)\s+last (\S+)', l)
        if m: d[k][m.group(1)] = float(m.group(2))
out = collections.OrderedDict()
for k, c in d.items():
    try:
        cu = c['SQ_BUSY_CU_CYCLES']
        out[k] = {'mfma_busy_frac': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * cu), 3),      # 4 SIMDs per CU
                  'valu_active_frac': round(4 * c['SQ_ACTIVE_INST_VALU'] / (4 * cu), 3),       # quad-cycle units
                  'wave_wait_any_frac': round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 3),
                  'wave_wait_inst_frac': round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 3),
                  'lds_bank_conflict_frac': round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), 3),
                  'mfma_coexec_frac_of_mfma_busy': round(c.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / max(c['SQ_VALU_MFMA_BUSY_CYCLES'], 1), 3),
                  'cu_cycles_per_cu': round(cu / 256), 'ms': c.get('duration_ms')}
    except KeyError:
        pass
sys.path.insert(0, os.environ['REPO_ROOT'] + '/tools')
from kernel_names import expected_kernels
missing = [k for k in expected_kernels(int(os.environ['PMC_W'])) if k not in out]
if missing:
    sys.exit('collect_profiles: kernels missing from the SQ counter passes: %s (names seen: %s)' % (missing, sorted(d)))
print(json.dumps({'lib_md5': os.environ.get('LIBMD5'), 'note': 'from sq_counters.txt (rocprofv3 --pmc SQ_* passes of tools/pmc_run.py, one launch each): mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) = fraction of SIMD cycles with the matrix pipe busy', 'kernels': out}, indent=1))
PY
done
need $O/${TAG}_sq_summary.json
need $O/${TAG}_w128_sq_summary.json
fi      # (TAIL_ONLY)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$R/bhnerf_amd/csrc -I$R/include $R/tools/step_bench.hip -o /tmp/step_bench > /tmp/step_bench.log 2>&1 || { tail -5 /tmp/step_bench.log >&2; exit 1; }
: > $O/${TAG}_telemetry_ceiling.txt
python3 $R/tools/smi_sample.py >> $O/${TAG}_telemetry_ceiling.txt & SMI=$!
sleep 2; /tmp/step_bench ceiling 8 > $O/${TAG}_ring_ceiling_microbench.txt 2>&1; sleep 1; kill $SMI || true
need $O/${TAG}_ring_ceiling_microbench.txt
: > $O/${TAG}_telemetry_bench.txt
python3 $R/tools/smi_sample.py >> $O/${TAG}_telemetry_bench.txt & SMI=$!
sleep 2; date +"# bench.py --steps 400 starts %s" >> $O/${TAG}_telemetry_bench.txt
python3 $R/bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-parity-mode --no-tape8 --no-tutorial-domain --no-other-configs --no-width128 > /tmp/bench400.json 2> /tmp/bench400.err
date +"# bench.py ends %s" >> $O/${TAG}_telemetry_bench.txt; sleep 1; kill $SMI || true
python3 -c "import json; d=json.load(open('/tmp/bench400.json')); print('# bench.py --steps 400: ms_per_step %.3f' % d['ms_per_step'])" >> $O/${TAG}_telemetry_bench.txt
ls -la $O
