#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
timeout 1200 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_fullsize_stokes.py -x -q -m gpu 2>&1 | tail -4
cat > /tmp/tw.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
geo = synthetic.synthetic_geodesics(128, 128, 64)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for width, depth, mode in ((128, 4, 'bf16'), (64, 4, 'bf16'), (64, 8, 'bf16'), (32, 4, 'bf16'), (64, 4, 'f32')):
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=depth, net_width=width, mode=mode, device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(depth, width).init(1, 21)))
    tM0 = engine.frame_offsets(np.linspace(0, 1, 8), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = torch.rand((8, 1, geom.R), device=dev) * 1e-3
    eng.render_train(geom, tM0)
    print(os.path.basename(os.environ.get('BHNERF_HIP_LIB', 'product')), '%dx%d %s: infer %.3f  fwd_train %.3f  bwd %.3f' % (depth, width, mode,
          timed(lambda: eng.render(geom, tM0)), timed(lambda: eng.render_train(geom, tM0)), timed(lambda: eng.render_bwd_tape(geom, tM0, dimg))))
PY
for r in 1 2; do for l in libbhnerf_hip_nores.so libbhnerf_hip.so; do BHNERF_HIP_LIB=$R/bhnerf_amd/csrc/$l timeout 200 python3 /tmp/tw.py 2>&1 | grep ":"; done; done | tee $O/resident_ab.txt
