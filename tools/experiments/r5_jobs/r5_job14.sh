#!/bin/bash
# round 5, job 14: new LDS swizzle of the fused width-128 backward: correctness, time, SQ_LDS_BANK_CONFLICT
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_job14; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize_stokes.py tests/test_gpu_api.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
for l in libbhnerf_hip_base.so libbhnerf_hip.so libbhnerf_hip_base.so libbhnerf_hip.so; do echo -n "$l "; BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/$l python bench.py --width 128 --steps 30 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 --no-tape8 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(round(d['ms_per_step'],3), [round(v,3) for v in k.values()])"; done | tee $O/ab_w128.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_c -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py bf16 128 > /tmp/pmc_c.log 2>&1
f=$(find /tmp/pmc_c -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/$O/lds_conflicts_w128.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if any(s in r['Kernel_Name'] for s in ('bwd128', 'chain_kernel', 'fused_fwd')): agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    c, a = d['SQ_LDS_BANK_CONFLICT'][-1], d['SQ_LDS_IDX_ACTIVE'][-1]
    print('%-72s SQ_LDS_BANK_CONFLICT %.4g  SQ_LDS_IDX_ACTIVE %.4g  frac %.4f' % (k, c, a, c / max(a, 1)))
PY
