"""Development aid: the 8-bit tape's state block (scales, maxima, ratios, |dimages|max) after each of a few backward calls
with differently scaled d(loss)/d(images), and the per-layer error of each call against the bf16 mode."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
geo = synthetic.synthetic_geodesics(16, 16, 64, seed=5)
out = {}
SLAB = 256 * 9 * 10 * 1024 * 4
for mode in ('bf16', 'bf16_t8'):
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=256, mode=mode, device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(4, 256).init(1, 21)))
    tM0 = engine.frame_offsets(np.linspace(0, 0.8, 2), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    gen = torch.Generator(device='cpu').manual_seed(7)
    dimg = ((torch.rand((2, 1, geom.R), generator=gen) - 0.3) * 1e-3).to(dev)
    res = []
    for factor in (1.0, 8.0, 1.0, 3.0):
        eng.render_train(geom, tM0)
        g = eng.render_bwd_tape(geom, tM0, factor * dimg).cpu() / factor
        res.append(g)
        if mode == 'bf16_t8':
            st = eng._ws[SLAB:SLAB + 128].view(torch.float32).cpu().numpy()
            print('factor %g: scales %s ratios %s dmax %.3e' % (factor, st[:4], st[16:20], st[24]))
    out[mode] = res
for i, (a, b) in enumerate(zip(out['bf16_t8'], out['bf16'])):
    ta, tb = eng.unflatten(a), eng.unflatten(b)
    per = ['%s.%s %.1e' % (ln[-1], pn[0], float((ta['MLP_0'][ln][pn] - tb['MLP_0'][ln][pn]).norm() / tb['MLP_0'][ln][pn].norm()))
           for ln in sorted(ta['MLP_0']) for pn in ('kernel', 'bias')]
    print(i, '%.2e' % float((a - b).norm() / b.norm()), ' '.join(per))
