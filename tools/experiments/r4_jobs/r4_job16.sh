#!/bin/bash
# dW kernel of the 8-bit tape mode under ablation builds (tools/build_variant.sh t8ablN "-DBHN_T8_ABL=N"): rocprofv3 average of dw_kernel<256, PolBF16T8>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4t8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for l in ${LIBS:-libbhnerf_hip.so}; do
  export BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l
  rm -rf /tmp/ka; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka -o kt -- python3 $R/tools/dbg_t8.py time > /tmp/ka.log 2>&1
  f=$(find /tmp/ka -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$l" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'dw_kernel' in n or ('chain_kernel' in n and 'T8' in n):
        out.append('%s %.0f' % ('dwT8' if 'dw_kernel' in n and 'T8' in n else 'dw16' if 'dw_kernel' in n else 'chn8' if ', 2,' in n else 'fwd8', float(r['AverageNs']) / 1e3))
print(sys.argv[2], ' '.join(sorted(out)))
PY
done | tee $O/ablation.txt
