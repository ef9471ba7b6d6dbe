"""Phase stamps of one workgroup iteration of the fused width-128 backward (build with -DBHN_B128_STAMPS=1:
bash tools/build_variant.sh st128 "-DBHN_B128_STAMPS=1"; BHNERF_HIP_LIB=.../libbhnerf_hip_st128.so python3 tools/dbg_bwd128_stamps.py)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=128, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, 128).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
names = ['iter start']
for l in (3, 2, 1):
    names += ['L%d setup' % l, 'L%d chain pb0 (+dma)' % l, 'L%d chain pb1 (+post0)' % l, 'L%d dW (+post1, w loads)' % l, 'L%d use_w' % l, 'L%d drain' % l, 'L%d barrier' % l]
names += ['L0 setup', 'L0 dW (+dma)', 'top phase of the next quad', 'drain', 'barrier']
for rep in range(3):
    eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg); torch.cuda.synchronize()
ws = eng._ws
ts = ws[(64 * 1024 + 256) * 4:(64 * 1024 + 256) * 4 + 4 * 64 * 8].view(torch.int64).cpu().numpy().reshape(4, 64)
for w in range(4):
    t = ts[w][:len(names)]
    print('wave %d: total %d cycles' % (w, t[-1] - t[0]))
    for i in range(1, len(names)):
        print('   %-22s +%6d' % (names[i], t[i] - t[i - 1]))
