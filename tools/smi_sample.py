#!/usr/bin/env python3
"""Board telemetry sampler for the measurement scripts: one line per sample
    <unix time> <socket power W> <mean GFX clock MHz over the XCDs> <min> <max> [<junction temperature C>]
from `amd-smi metric -p -c -t` (works as an ordinary user on the GPU box).  Runs until killed:
    python3 tools/smi_sample.py > gpurun_out/telemetry.txt &
Note (MI355X_MICROARCH.md, DVFS give-back 6): board power and the SMI clock are context, not the test -- the in-kernel clock
(s_memtime / s_memrealtime) printed by the micro-benchmarks is what the kernels actually ran at."""
import re
import subprocess
import sys
import time

while True:
    t = time.time()
    try:
        out = subprocess.run(['amd-smi', 'metric', '-p', '-c', '-t'], capture_output=True, text=True, timeout=10).stdout
    except Exception as exc:                     # noqa: BLE001
        print('# amd-smi failed: %s' % exc, flush=True)
        time.sleep(1.0)
        continue
    pw = re.search(r'SOCKET_POWER:\s*([\d.]+)', out)
    clks = [float(m) for m in re.findall(r'GFX_\d+:\s*\n\s*CLK:\s*([\d.]+)', out)]
    temp = re.search(r'(?:HOTSPOT|JUNCTION)[A-Z_]*:\s*([\d.]+)', out)
    if pw and clks:
        print('%.2f %s %.0f %.0f %.0f %s' % (t, pw.group(1), sum(clks) / len(clks), min(clks), max(clks), temp.group(1) if temp else '-'), flush=True)
    time.sleep(max(0.0, 0.2 - (time.time() - t)))
