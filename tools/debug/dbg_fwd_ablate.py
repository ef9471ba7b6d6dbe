"""Inference forward (4x256 bf16, config 2) with one cost knocked out at a time (fused_fwd_kernel DBG build)."""
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
width, H, G, B = 256, 128, 64, 8
geo = synthetic.synthetic_geodesics(H, H, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=width, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, width).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
names = {0: 'full (debug build)', 1: '- MFMAs', 2: '- relu/pack', 4: '- weight DMA', 8: '- barriers', 16: '- posenc trig', 32: '- epilogue',
         12: '- DMA - barriers', 3: '- MFMA - pack', 63: 'nothing left', 62: 'MFMAs only', 60: 'MFMA + pack only', 48: '- prologue - epilogue'}
lib.bhn_debug_set_fwd_variant(1)
print('%-24s %.3f ms' % ('production', timed(lambda: eng.render(geom, tM0))))
lib.bhn_debug_set_fwd_variant(1 | (64 << 4))
print('%-24s %.3f ms' % ('production, no phase lag', timed(lambda: eng.render(geom, tM0))))
import os
sel = names.items() if os.environ.get('ABLATE') else [(f, names[f]) for f in (0, 16, 32, 48, 1, 2)]
for f, n in sel:
    t = []
    for nolag in (0, 64):
        lib.bhn_debug_set_fwd_variant(3 | ((f | nolag) << 4))
        t.append(timed(lambda: eng.render(geom, tM0)))
    print('%-24s lag %.3f ms   no lag %.3f ms' % (n, t[0], t[1]))
lib.bhn_debug_set_fwd_variant(1)
