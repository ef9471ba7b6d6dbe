#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
timeout 300 python3 tools/dbg_grad_ref.py check 2>&1 | grep "w128\|OK\|FAIL" | tee $O/check.log
cat > /tmp/t128.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
geo = synthetic.synthetic_geodesics(128, 128, 64)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=128, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, 128).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, 8), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((8, 1, geom.R), device=dev) * 1e-3
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
eng.render_train(geom, tM0)
print(os.path.basename(os.environ.get('BHNERF_HIP_LIB', 'product')), 'fwd_train %.3f  bwd %.3f  infer %.3f' % (timed(lambda: eng.render_train(geom, tM0)), timed(lambda: eng.render_bwd_tape(geom, tM0, dimg)), timed(lambda: eng.render(geom, tM0))))
PY
for r in 1 2; do for l in ${LIBS:-libbhnerf_hip.so}; do
  BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l timeout 120 python3 /tmp/t128.py 2>&1 | tail -1
done; done | tee $O/time.txt
