import sys, numpy as np, torch, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0'); lib = _hip.lib()
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=256, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, 256).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, 64)[:B], 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for rnd in range(3):
    lib.bhn_debug_set_bwd_stages(7 | (2048 << 3))          # prologue computed
    img0 = eng.render_train(geom, tM0).clone(); g0 = eng.render_bwd_tape(geom, tM0, dimg).clone()
    t_c = timed(lambda: eng.render_train(geom, tM0))
    lib.bhn_debug_set_bwd_stages(7)                         # encoded inputs loaded from the tape of the previous launch
    img1 = eng.render_train(geom, tM0).clone(); g1 = eng.render_bwd_tape(geom, tM0, dimg).clone()
    t_l = timed(lambda: eng.render_train(geom, tM0))
    print('round %d: training forward  computed %.3f ms   loaded %.3f ms   images equal %s  grads equal %s' % (rnd, t_c, t_l, torch.equal(img0, img1), torch.equal(g0, g1)))
