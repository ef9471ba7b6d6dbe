"""A/B of the inference forward (render, predict) at config 2's shape (128x128 rays x 64 samples, 8 frames) per library variant:
   BHNERF_HIP_LIB=<lib> python3 tools/ab_infer.py [width depth]   (run per variant, interleaved by the calling job)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
geo = synthetic.synthetic_geodesics(128, 128, 64)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=depth, net_width=width, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(depth, width).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, 8), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
img = eng.render(geom, tM0)
print(os.path.basename(os.environ.get('BHNERF_HIP_LIB', 'product')), '%dx%d render %.3f ms  predict %.3f ms  image sum %.6e' % (
    depth, width, timed(lambda: eng.render(geom, tM0)), timed(lambda: eng.predict(geom, tM0)), float(img.double().sum())))
