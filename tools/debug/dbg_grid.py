"""GRID_Predictor kernels at config-2 size (128x128 rays x 64 samples, 8 frames, 64^3 grid): render and gradient time."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.GRID_Predictor(8.0, 2.0, 8.0, 4.0, 64, device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
flat = (torch.rand(64 ** 3, device=dev) * 6 + 6).contiguous()
eng.pack(flat)
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((B, 1, geom.R), device=dev)
pts = B * geom.P
t_f = timed(lambda: eng.render(geom, tM0)); t_b = timed(lambda: eng.render_bwd(geom, tM0, dimg))
act = float(geom.active_fraction)
print('render %.3f ms (%.1f G points/s, geometry stream %.0f GB/s)   gradient %.3f ms (%.1f G points/s, %.2f G atomics/s on %.0f %% in-domain points)' % (
    t_f, pts / t_f / 1e6, pts * 25 / t_f / 1e6, t_b, pts / t_b / 1e6, pts * act * 8 / t_b / 1e6, 100 * act))
