"""Round 6: the 4x128 inference forward (resident weights, 12-wave workgroups, config 2's geometry) with one cost knocked out at
a time -- fused_fwd_kernel<128, PolBF16X, .., DBG, RES> of the debug build (make -C bhnerf_amd/csrc debug).  Results of the
ablated launches are meaningless; the times say what the kernel is made of."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('BHNERF_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bhnerf_amd', 'csrc', 'libbhnerf_hip_dbg.so'))
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


lib = _hip.lib()
H, G, B = 128, 64, 8
geo = synthetic.synthetic_geodesics(H, H, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=128, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, 128).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
names = [(0, 'full (debug build)'), (16, '- prologue (warp, trig, fragments)'), (32, '- epilogue (sigmoid, ray sums)'), (48, '- prologue - epilogue'),
         (1, '- MFMAs'), (2, '- relu / pack'), (3, '- MFMAs - relu / pack'), (51, 'loads + ring bookkeeping only'), (50, 'MFMAs only (+ loads)')]
lib.bhn_debug_set_fwd_variant(1)
print('%-40s %.3f ms' % ('production kernel', timed(lambda: eng.render(geom, tM0))))
for f, n in names:
    lib.bhn_debug_set_fwd_variant(3 | (f << 4))
    print('%-40s %.3f ms' % (n, timed(lambda: eng.render(geom, tM0))))
lib.bhn_debug_set_fwd_variant(1)
