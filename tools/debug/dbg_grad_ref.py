"""Development aid: dump the bf16 parameter gradients of a set of small problems (`save`), or compare the current
build against a dump (`check`, file tools/_grad_ref.npz) -- used when the tape layout or the dW jobs change, to
separate logic errors (large differences) from summation-order changes (~1e-6)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bhnerf_amd import engine, network, synthetic, constants

dev = torch.device('cuda:0')
CASES = [  # name, H, W, G, B, width, depth, do_skip, S, masked
    ('w256d4', 24, 24, 64, 3, 256, 4, True, 0, False),
    ('w256d4_stokes_masked', 16, 16, 100, 2, 256, 4, True, 3, True),
    ('w128d4', 16, 16, 64, 3, 128, 4, True, 0, True),
    ('w64d8', 16, 16, 48, 2, 64, 8, True, 2, False),
    ('w32d6', 8, 8, 40, 2, 32, 6, True, 0, False),
    ('w256d3_noskip', 16, 16, 64, 2, 256, 3, False, 0, False),
    ('w128d5_noskip', 16, 16, 64, 2, 128, 5, False, 0, True),
    ('w256d2_noskip', 16, 16, 64, 2, 256, 2, False, 0, False),
]


ENG = {}


def grads():
    out = {}
    for name, H, W, G, B, width, depth, skip, S, masked in CASES:
        geo = synthetic.synthetic_geodesics(H, W, G, S=S, seed=5)
        dom = (8.0, 2.0, 8.0, 4.0) if masked else (8.0, 0.0, np.inf, np.inf)
        pred = network.NeRF_Predictor(*dom, net_depth=depth, net_width=width, do_skip=skip, mode='bf16', device=dev)
        eng = pred.engine()
        geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], geo['J'] if S else None, geo['g'], geo['dtau'], geo['Sigma'])
        flat = eng.flatten(network.MLP(depth, width, do_skip=skip).init(1, 21))
        gen = torch.Generator(device='cpu').manual_seed(7)
        flat = flat + 0.02 * torch.randn(flat.shape, generator=gen).to(flat)          # non-zero biases
        eng.pack(flat)
        tM0 = engine.frame_offsets(np.linspace(0, 0.8, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
        dimg = (torch.rand((B, max(S, 1), geom.R), generator=gen) - 0.3).to(dev)
        out[name] = eng.render_bwd(geom, tM0, dimg).cpu().numpy().copy()
        ENG[name] = eng
        if eng.fits_tape(B, geom.P_eff):
            eng.render_train(geom, tM0)
            out[name + '_taped'] = eng.render_bwd_tape(geom, tM0, dimg).cpu().numpy().copy()
    return out


if sys.argv[1] == 'save':
    os.makedirs('gpurun_out', exist_ok=True)
    np.savez('gpurun_out/grad_ref.npz', **grads())
    print('saved')
else:
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_grad_ref.npz'))
    bad = 0
    for k, g in grads().items():
        r = ref[k]
        err = np.abs(g - r).max() / np.abs(r).max()
        l2 = np.linalg.norm(g - r) / np.linalg.norm(r)
        flag = '' if err < 2e-5 else '   <-- DIFFERS'
        bad += err >= 2e-5
        print('%-28s max err %.2e  L2 %.2e  |g|max %.3e%s' % (k, err, l2, np.abs(r).max(), flag))
        if err >= 2e-5 and k in ENG:
            tg, tr = ENG[k].unflatten(torch.as_tensor(g)), ENG[k].unflatten(torch.as_tensor(r))
            for lname in sorted(tg['MLP_0']):
                for pn in ('kernel', 'bias'):
                    a, b = tg['MLP_0'][lname][pn].numpy(), tr['MLP_0'][lname][pn].numpy()
                    print('      %s.%s  max err %.2e of %.2e' % (lname, pn, np.abs(a - b).max(), np.abs(b).max()))
                    if np.abs(a - b).max() > 1e-4 * np.abs(b).max() and a.shape[-1] == 1:
                        e = np.abs(a - b).reshape(-1, 32).max(1)
                        print('         per 32-block:', ' '.join('%.1e' % v for v in e))
                        print('         got ', a.ravel()[:8], a.ravel()[32:40])
                        print('         want', b.ravel()[:8], b.ravel()[32:40])
    print('FAIL' if bad else 'OK')
