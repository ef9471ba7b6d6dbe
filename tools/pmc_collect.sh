#!/bin/bash
# rocprofv3 SQ counter passes over tools/pmc_run.py; summaries -> gpurun_out/pmc/pass*.txt
#   bash tools/pmc_collect.sh [bf16|f32] [width]
MODE=${1:-bf16}
WIDTH=${2:-256}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR" \
           "SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python3 $R/tools/pmc_run.py $MODE $WIDTH > /tmp/pmc_$i.log 2>&1
  grep -i -E "error|traceback|exception" /tmp/pmc_$i.log | head -5; ls -R /tmp/pmc_$i | head -8; f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  echo "pass $i: $f"
  [ -n "$f" ] && python3 - "$f" $R > $R/gpurun_out/pmc/pass$i.txt <<'PY'
import csv, sys, collections
sys.path.insert(0, sys.argv[2] + '/tools')
from kernel_names import short_name, FWD_NAME, CHAIN_NAME, FUSED_NAME, INFER_NAME
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = short_name(r['Kernel_Name'])                 # (template arguments matched by position: tools/kernel_names.py)
    if k not in (FWD_NAME, CHAIN_NAME, FUSED_NAME, INFER_NAME, 'dw_kernel'): continue
    d = agg.setdefault(k, collections.OrderedDict())
    d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    d.setdefault('duration_ms', []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print('   %-34s last %.4g  (n=%d)' % (c, v[-1], len(v)))
PY
done
cat $R/gpurun_out/pmc/pass*.txt
