import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import os; os.environ.setdefault('BHNERF_HIP_LIB', '/root/repo/bhnerf_amd/csrc/libbhnerf_hip_dbg.so')   # debug build: make -C bhnerf_amd/csrc debug
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')
MODE = sys.argv[1] if len(sys.argv) > 1 else 'bf16'       # python tools/dbg_dw.py [bf16|f32]
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=256, mode=MODE, device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
flat = eng.flatten(network.MLP(4, 256).init(1, 21)); eng.pack(flat)
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
lib.bhn_debug_set_bwd_stages(7); eng.render_bwd(geom, tM0, dimg)
cases = [('chain', 1), ('dw full', 2), ('dw no-mfma', 2 | 8), ('dw no-loads', 2 | 16)]
for j in range(5):
    cases += [('job%d full' % j, 2 | ((j + 1) << 5)), ('job%d no-loads' % j, 2 | 16 | ((j + 1) << 5)), ('job%d no-mfma' % j, 2 | 8 | ((j + 1) << 5))]
for name, mask in cases:
    lib.bhn_debug_set_bwd_stages(mask)
    print(name, '%.3f ms' % timed(lambda: eng.render_bwd(geom, tM0, dimg)))
