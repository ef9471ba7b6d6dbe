"""Development aid for the 8-bit tape mode (BHN_BF16_T8): gradient of `bf16_t8` against `bf16` on small 4x256 problems, layer
by layer (expected: ~2e-3 of each layer's norm, tools/exp_fp8_tape_accuracy.py), on the calibrating first call and on the
second call (scales from the first); then the kernel times of both modes at BASELINE config 2's shape.
    python tools/dbg_t8.py [check|time|all]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bhnerf_amd import engine, network, synthetic, constants

dev = torch.device('cuda:0')
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
CASES = [  # name, H, W, G, B, depth, do_skip, S, masked
    ('w256d4', 24, 24, 64, 3, 4, True, 0, False),
    ('w256d4_stokes_masked', 16, 16, 100, 2, 4, True, 3, True),
    ('w256d3_noskip', 16, 16, 64, 2, 3, False, 0, False),
    ('w256d6', 16, 16, 64, 2, 6, True, 0, False),
]


def build(mode, H, W, G, B, depth, skip, S, masked, seed=5):
    geo = synthetic.synthetic_geodesics(H, W, G, S=S, seed=seed)
    dom = (8.0, 2.0, 8.0, 4.0) if masked else (8.0, 0.0, np.inf, np.inf)
    pred = network.NeRF_Predictor(*dom, net_depth=depth, net_width=256, do_skip=skip, mode=mode, device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], geo['J'] if S else None, geo['g'], geo['dtau'], geo['Sigma'])
    flat = eng.flatten(network.MLP(depth, 256, do_skip=skip).init(1, 21))
    gen = torch.Generator(device='cpu').manual_seed(7)
    flat = flat + 0.02 * torch.randn(flat.shape, generator=gen).to(flat)
    eng.pack(flat)
    tM0 = engine.frame_offsets(np.linspace(0, 0.8, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = ((torch.rand((B, max(S, 1), geom.R), generator=gen) - 0.3) * 1e-3).to(dev)
    return eng, geom, tM0, dimg


bad = 0
if what in ('check', 'all'):
    for name, *cfg in CASES:
        g = {}
        for mode in ('bf16', 'bf16_t8'):
            eng, geom, tM0, dimg = build(mode, *cfg)
            img = eng.render_train(geom, tM0)
            g[mode] = eng.render_bwd_tape(geom, tM0, dimg).cpu()
            if mode == 'bf16_t8':
                eng.render_train(geom, tM0)
                g['second'] = eng.render_bwd_tape(geom, tM0, 3.0 * dimg).cpu() / 3.0       # scales from the first call, 3x the gradient
                g['recompute'] = eng.render_bwd(geom, tM0, dimg).cpu()
        ref = g['bf16']
        print('%s: |g| %.3e' % (name, float(ref.norm())))
        for k in ('bf16_t8', 'second', 'recompute'):
            tot = float((g[k] - ref).norm() / ref.norm())
            tg, tr = eng.unflatten(g[k]), eng.unflatten(ref)
            per = []
            for lname in sorted(tg['MLP_0'], key=lambda s: int(s.split('_')[1])):
                for pn in ('kernel', 'bias'):
                    a, b = tg['MLP_0'][lname][pn], tr['MLP_0'][lname][pn]
                    per.append('%s.%s %.1e' % (lname.replace('Dense_', 'L'), pn[0], float((a - b).norm() / max(float(b.norm()), 1e-30))))
            ok = np.isfinite(tot) and tot < 1e-2
            bad += not ok
            print('   %-10s rel L2 %.2e %s   %s' % (k, tot, '' if ok else '<-- BAD', '  '.join(per)))
    print('FAIL' if bad else 'OK')

if what in ('time', 'all'):
    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev]))
    for rep in range(2):
        for mode in ('bf16', 'bf16_t8'):
            eng, geom, tM0, dimg = build(mode, 128, 128, 64, 8, 4, True, 0, False, seed=0)
            eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg)
            print('%-8s fwd_train %.3f  bwd %.3f ms' % (mode, timed(lambda: eng.render_train(geom, tM0)), timed(lambda: eng.render_bwd_tape(geom, tM0, dimg))))
sys.exit(1 if bad else 0)
