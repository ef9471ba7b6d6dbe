#!/bin/bash
# Round 6: every GPU job of the round, one function per job.   gpurun -- bash tools/experiments/r6_jobs.sh <job> [args]
# Output under gpurun_out/r6_<job>/ (scratch); what is judged is copied into profiles/r6_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; J=$1; shift
O=$R/gpurun_out/r6_$J; mkdir -p $O
C=$R/bhnerf_amd/csrc
cd $R

# A/B of library variants on this box: step + kernel times, two interleaved rounds (tools/ab.sh)
ab() { bash tools/ab.sh "$@" 2>&1 | grep -v amdgpu.ids; }
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
*) echo "unknown job $J"; exit 1;;
esac
