#!/bin/bash
# Round 6: every GPU job of the round, one function per job.   gpurun -- bash tools/experiments/r6_jobs.sh <job> [args]
# Output under gpurun_out/r6_<job>/ (scratch); what is judged is copied into profiles/r6_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; J=$1; shift
O=$R/gpurun_out/r6_$J; mkdir -p $O
C=$R/bhnerf_amd/csrc
cd $R

# A/B of library variants on this box: step + kernel times, two interleaved rounds (tools/ab.sh)
ab() { bash tools/ab.sh "$@" 2>&1 | grep -v amdgpu.ids; }
# the same with the 4x128 block of bench.py: step, [training forward, fused backward], inference, of both networks
ab128() {
  for r in 1 2; do for l in "$@"; do echo -n "$l "; BHNERF_HIP_LIB=$C/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-tape8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; w=d['width128']; rw=w['roofline']
print('4x256 step %.3f' % d['ms_per_step'], [round(v,3) for v in r['kernel_ms'].values()], '| 4x128 step %.3f graph %.3f' % (w['ms_per_step'], w.get('ms_per_step_hip_graph', 0)), [round(v,3) for v in rw['kernel_ms'].values()])"; done; done
}
# one SQ counter pass (group 3: LDS) over the step's kernels for library $1, width $2
lds_pass() {
  local lib=$1 width=${2:-256} tag=$3
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag &&
    BHNERF_HIP_LIB=$C/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES \
      --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python3 $R/tools/pmc_run.py bf16 $width > /tmp/pmc_$tag.log 2>&1
    f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" $R <<'PY'
import csv, sys, collections
sys.path.insert(0, sys.argv[2] + '/tools')
from kernel_names import short_name
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = short_name(r['Kernel_Name'])
    if not k or 'kernel' not in k: continue
    agg.setdefault(k, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, d in agg.items():
    if not any(s in k for s in ('chain', 'dw_', 'fused', 'bwd128')): continue
    last = {c: v[-1] for c, v in d.items()}
    print('%-40s' % k, ' '.join('%s=%.4g' % (c.replace('SQ_', ''), v) for c, v in last.items()),
          'conflict/idx_active=%.4f' % (last.get('SQ_LDS_BANK_CONFLICT', 0) / max(last.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
  )
}

case $J in
swz)        # experiment (b): swizzled gA_0 staging images in chain_kernel<CHAIN> -- parity, A/B, LDS counters
  python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; tail -3 $O/pytest.txt
  ab libbhnerf_hip_swz0.so libbhnerf_hip.so | tee $O/ab.txt
  for l in libbhnerf_hip_swz0.so libbhnerf_hip.so; do echo "== $l"; lds_pass $l 256 ${l%.so}; done | tee $O/lds.txt
  ;;
validate)   # ABI 5 build: whole GPU suite, smoke, the default bench line (every block), LDS counters of the swizzle A/B
  python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest.txt; tail -25 $O/pytest.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/smoke.txt
  python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('step %.3f ms  value %.4g  roofline %s frac %.4f clock %s' % (d['ms_per_step'], d['value'], r['kernel'], r['frac'], r.get('sustained_clock_mhz')))
for k, v in r['kernels'].items(): print('  %-32s %.3f ms  %d flop/pt  mfma %.4f  tape %.0f B/pt  %.0f GB/s  clock %s' % (k, v['ms'], v['flop_per_point'], v['mfma_frac'], v['tape_bytes_per_point'], v['tape_GB_per_s'], v['sustained_clock_mhz']))
print('  kernel_ms_sum %.3f  inference %s  peak_this_box %s' % (r['kernel_ms_sum'], r['inference_forward'], r.get('mfma_peak_this_box')))
w = d.get('width128', {})
print('width128 step', w.get('ms_per_step'), 'graph', w.get('ms_per_step_hip_graph'), {k: (v['ms'], v['mfma_frac'], v['sustained_clock_mhz']) for k, v in w.get('roofline', {}).get('kernels', {}).items()} if 'roofline' in w else w)
print('fwd images/s', d.get('fwd_images_per_s'), 'cpu', {k: d['cpu_baseline'][k] for k in ('value', 'cores', 'fwd_images_per_s')} if d.get('cpu_baseline') else None)
print('general_path', d.get('general_path', {}).get('ms_per_step'), 'other', {k: v.get('ms_per_step') for k, v in d.get('other_configs', {}).items() if isinstance(v, dict)})
PY
  for l in libbhnerf_hip.so; do echo "== $l"; lds_pass $l 256 ${l%.so}; done | tee $O/lds.txt
  ;;
pairs)      # item 3: paired (sin, cos) slot layout of the encoded inputs -- parity, A/B on both networks, ablation of the 4x128 forward
  python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt; tail -12 $O/pytest.txt
  ab128 libbhnerf_hip_pairs0.so libbhnerf_hip.so | tee $O/ab.txt
  python tools/dbg_fwd128_ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/ablate128.txt
  ;;
flags)      # experiment (c): per-slot FULL / FREE words instead of the per-chunk barrier, 4x256 inference forward
  BHNERF_HIP_LIB=$C/libbhnerf_hip_flags.so timeout 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
  for r in 1 2 3; do for l in libbhnerf_hip.so libbhnerf_hip_flags.so; do BHNERF_HIP_LIB=$C/$l timeout 300 python tools/ab_infer.py 256 4 2>&1 | tail -1; done; done | tee $O/ab.txt
  for l in libbhnerf_hip.so libbhnerf_hip_flags.so; do BHNERF_HIP_LIB=$C/$l timeout 300 python tools/ab_infer.py 256 8 2>&1 | tail -1; done | tee -a $O/ab.txt
  for l in libbhnerf_hip.so libbhnerf_hip_flags.so; do echo "== $l"; lds_pass $l 256 ${l%.so}; done | grep -E "==|inference" | tee $O/sq.txt
  ;;
encw)       # resident-weights kernels stream KS fragments per chunk (encoded-input pair read in place): parity + A/B at 4x128
  python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt; tail -4 $O/pytest.txt
  ab128 libbhnerf_hip_encw0.so libbhnerf_hip.so | tee $O/ab.txt
  ;;
dwub)       # experiment (a), upper bound first: the dW kernel with the LDS traffic of 4 x 4 register blocking (every second B fragment
            # not read: results wrong), with the LBITS / DROP_HD byte cut, and with both; + the general path's tests and step time
  python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "general or outside" 2>&1 | tail -12 | tee $O/pytest.txt
  ab libbhnerf_hip.so libbhnerf_hip_rd.so libbhnerf_hip_lb.so libbhnerf_hip_lbrd.so | tee $O/ab.txt
  python tools/general_path_bench.py 8 2 2>&1 | grep -v amdgpu | tail -8 | tee $O/general.txt
  ;;
gen16)      # general path in bf16 (gen_mlp16_kernel / gen_dw16_kernel): parity and step times
  timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -s -k "general or outside" 2>&1 | grep -v Warning | tail -40 | tee $O/pytest.txt
  timeout 900 python tools/general_path_bench.py 8 2 2>&1 | grep -v amdgpu | tail -12 | tee $O/general.txt
  ;;
genprof)    # which kernel of the bf16 general path dominates (4x128 deg 5: GEN_ONLY=1; 4x512: GEN_ONLY=3)
  for only in 1 3; do
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kg && GEN_ONLY=$only rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kg -o k -- python3 $R/tools/general_path_bench.py 8 3 > $O/run$only.txt 2>&1
      f=$(find /tmp/kg -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:8]: print('%-90s calls %5s  avg %10.1f us  %5s %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
    ); grep general $O/run$only.txt
  done | tee $O/stats.txt
  ;;
fuzz)       # randomised parity sweeps on the final library: fused shapes, general shapes, fused-vs-generic 4x128 backward, stress
  timeout 1500 python tools/fuzz_parity.py ${1:-300} 6 2>&1 | grep -v amdgpu | tail -40 | tee $O/fuzz.txt
  FUZZ_GENERAL=1 timeout 900 python tools/fuzz_parity.py ${2:-60} 7 2>&1 | grep -v amdgpu | tail -20 | tee $O/fuzz_general.txt
  BHNERF_HIP_LIB=$C/libbhnerf_hip_nof128.so timeout 900 python tools/fuzz_fused128.py save 100 8 2>&1 | grep -v amdgpu | tail -2
  timeout 900 python tools/fuzz_fused128.py check 100 8 2>&1 | grep -v amdgpu | tail -5 | tee $O/fuzz_fused128.txt
  timeout 600 python tools/debug/dbg_stress.py 2>&1 | grep -v amdgpu | tail -8 | tee $O/stress.txt
  ;;
fuzzold)    # the HARD draws of the round-6 sweep (bf16, deep narrow networks on tiny ray sets) on round 5's library, built from a
            # tree of commit 89fdb6c under tools/_r5tree (git archive 89fdb6c | tar -x -C tools/_r5tree; make -C .../csrc): regression or statistics?
  echo "== round-6 library"; FUZZ_ONLY=$1 timeout 900 python tools/fuzz_parity.py 1000 6 2>&1 | grep -E "draw|random conf" | tee $O/new.txt
  echo "== round-5 library (commit 89fdb6c)"; ( cd tools/_r5tree && FUZZ_ONLY=$1 timeout 900 python tools/fuzz_parity.py 1000 6 2>&1 | grep -E "draw|random conf" ) | tee $O/old.txt
  ;;
final)      # everything under profiles/r6_* that comes from tools/collect_profiles.sh, on the final library; then the two examples
  bash tools/collect_profiles.sh r6 2>&1 | tail -15
  timeout 600 python examples/image_plane_recovery.py 2>&1 | grep -v amdgpu | tail -4 | tee $O/example_recovery.txt
  ;;
ceiling)    # the tail of tools/collect_profiles.sh alone (ring-ceiling micro-benchmark + telemetry; round 6: tools/step_bench.hip did not compile in the first collection)
  TAIL_ONLY=1 bash tools/collect_profiles.sh r6 2>&1 | tail -8
  ;;
spread)     # one default bench.py run on whatever box this call gets: the box-to-box spread of the final library (profiles/r6_box_spread.txt)
  python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; w=d['width128']; rw=w['roofline']; pk=r['mfma_peak_this_box']
print('4x256 step %.3f ms frac %.4f | fwd %.3f chain %.3f dW %.3f | clocks %s | infer %.3f (%.3f) | peak_this_box %.0f TF @ %.0f MHz | 4x128 step %.3f [%.3f %.3f] infer %.3f | images/s %.0f | general %.1f ms | cfg3 %.2f cfg5 %.3f' % (
  d['ms_per_step'], r['step_mfma_frac'], r['kernel_ms'][list(r['kernels'])[0]], r['kernel_ms'][list(r['kernels'])[1]], r['kernel_ms'][list(r['kernels'])[2]],
  [int(v['sustained_clock_mhz']) for v in r['kernels'].values()], r['inference_forward']['ms'], r['inference_forward']['mfma_frac'], pk['tflops'], pk['clock_mhz'],
  w['ms_per_step'], *[v['ms'] for v in rw['kernels'].values()], rw['inference_forward']['ms'], d['fwd_images_per_s'], d['general_path']['ms_per_step'],
  d['other_configs']['config3']['ms_per_step'], d['other_configs']['config5']['ms_per_step']))" | tee $O/line_$(date +%s).txt
  ;;
examples)   # the two end-to-end examples on the final library
  timeout 600 python examples/image_plane_recovery.py 2>&1 | grep -v amdgpu | tail -3 | tee $O/recovery.txt
  ( cd /tmp && timeout 900 python $R/examples/fit_alma_lp.py 20 30 40 --config $R/examples/fit_alma_lp.yaml 2>&1 | grep -v amdgpu | tail -8 ) | tee $O/alma.txt
  ;;
sweep)      # re-sweep of compile-time tunables after this round's changes: libbhnerf_hip_<name>.so variants given as arguments, against the product
  ab libbhnerf_hip.so "$@" | tee $O/ab_$(date +%s).txt
  ;;
sweep128)   # the same for the 4x128 path (bench.py with its width-128 block): variants against the product
  ab128 libbhnerf_hip.so "$@" | tee $O/ab_$(date +%s).txt
  ;;
genab)      # general bf16 path: step time of 4x128 deg 5 / 4x256 deg 6 / 4x512 per library variant (the product library = the committed one)
  BHNERF_HIP_LIB=$C/$2 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "general or outside" 2>&1 | tail -3 | tee $O/pytest.txt      # (parity of the second library given)
  for r in 1 2; do for l in "$@"; do for only in 1 2 3; do echo -n "$l "; BHNERF_HIP_LIB=$C/$l GEN_ONLY=$only python tools/general_path_bench.py 8 3 2>&1 | grep general; done; done; done | tee $O/ab.txt
  ;;
*) echo "unknown job $J"; exit 1;;
esac
