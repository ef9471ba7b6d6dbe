import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import os; os.environ.setdefault('BHNERF_HIP_LIB', '/root/repo/bhnerf_amd/csrc/libbhnerf_hip_dbg.so')   # debug build: make -C bhnerf_amd/csrc debug
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
lib = _hip.lib()
for width, H in ((256, 128), (128, 128)):
    G = 64; B = 8
    geo = synthetic.synthetic_geodesics(H, H, G)
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=width, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(4, width).init(1, 21)))
    tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    res = {}
    for v in (1,):
        lib.bhn_debug_set_fwd_variant(v)
        img = eng.render(geom, tM0).clone()
        res[v] = (timed(lambda: eng.render(geom, tM0)), img)
    flops = 2 * (21 * width + 2 * width * width + (width + 21) * width + width) * B * geom.P
    print('width %d: inference forward %.3f ms (%.0f TF/s)' % (width, res[1][0], flops / res[1][0] / 1e9))
lib.bhn_debug_set_fwd_variant(1)
