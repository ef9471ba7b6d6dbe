"""One training-forward, delta-chain, dW and inference-forward launch at config 2 (for rocprofv3 --pmc passes).
    python3 tools/pmc_run.py [bf16|f32] [width]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')
MODE = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
WIDTH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=WIDTH, mode=MODE, device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
flat = eng.flatten(network.MLP(4, WIDTH).init(1, 21)); eng.pack(flat)
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
for _ in range(2):
    eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg); eng.render(geom, tM0)
torch.cuda.synchronize()
