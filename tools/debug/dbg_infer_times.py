import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for width, depth, S, G in ((128, 4, 0, 64), (128, 4, 3, 100), (64, 4, 0, 64), (64, 8, 0, 64), (32, 6, 0, 64)):
    geo = synthetic.synthetic_geodesics(128, 128, G, S=S)
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=depth, net_width=width, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], geo['J'] if S else None, geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(depth, width).init(1, 21)))
    tM0 = engine.frame_offsets(np.linspace(0, 1, 8), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    print(os.path.basename(os.environ.get('BHNERF_HIP_LIB', 'product')), '%dx%d S=%d G=%d render %.3f ms  predict %.3f ms' % (depth, width, S, G, timed(lambda: eng.render(geom, tM0)), timed(lambda: eng.predict(geom, tM0))))
