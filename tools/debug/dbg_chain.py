import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import os; os.environ.setdefault('BHNERF_HIP_LIB', '/root/repo/bhnerf_amd/csrc/libbhnerf_hip_dbg.so')   # debug build: make -C bhnerf_amd/csrc debug
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=int(os.environ.get('WIDTH', 256)), mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
flat = eng.flatten(network.MLP(4, int(os.environ.get('WIDTH', 256))).init(1, 21)); eng.pack(flat)
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))
lib = _hip.lib()
eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg)
print('inference fwd %.3f ms' % timed(lambda: eng.render(geom, tM0)))
lib.bhn_debug_set_bwd_stages(1)
print('bhn_render_bwd, forward + chain stage %.3f ms' % timed(lambda: eng.render_bwd(geom, tM0, dimg)))
lib.bhn_debug_set_bwd_stages(7)
for name, bits in (('full', 0), ('emit w/o global stores', 1 << 9), ('no emit', 1 << 10)):
    lib.bhn_debug_set_bwd_stages(7 | bits)
    t1 = timed(lambda: eng.render_train(geom, tM0))
    lib.bhn_debug_set_bwd_stages(1 | bits)
    t2 = timed(lambda: eng.render_bwd_tape(geom, tM0, dimg))
    print('%-24s fwd_train %.3f ms   chain %.3f ms' % (name, t1, t2))
lib.bhn_debug_set_bwd_stages(7)
