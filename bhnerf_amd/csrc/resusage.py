#!/usr/bin/env python3
"""Print a per-kernel register/LDS/scratch table: hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'remark: (?:[^:]*:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
    if not m:
        continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(' [')[0]] = v
print('%-70s %5s %5s %5s %7s %4s %6s' % ('kernel', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'occ', 'vspill'))
for r in rows:
    print('%-70s %5s %5s %5s %7s %4s %6s' % (r['name'][:70], r.get('VGPRs'), r.get('AGPRs'), r.get('SGPRs'),
                                            r.get('ScratchSize'), r.get('Occupancy'), r.get('VGPRs Spill')))
