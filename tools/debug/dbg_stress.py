"""Stress the counted-vmcnt rings: config-2 size (and a ragged 100-sample geometry), many repetitions of the taped step
with and without a concurrent copy stream hammering HBM; every gradient must equal the first bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bhnerf_amd import engine, network, synthetic, constants
dev = torch.device('cuda:0')
big = torch.empty(2 * 2**30 // 4, device=dev); big2 = torch.empty_like(big)
side = torch.cuda.Stream()
for (H, G, width, depth, S, mode) in [(128, 64, 256, 4, 0, 'bf16'), (64, 100, 128, 4, 3, 'bf16'), (96, 64, 256, 8, 0, 'bf16'), (64, 48, 64, 6, 2, 'bf16'),
                                     (128, 64, 256, 4, 0, 'bf16_t8'), (64, 100, 256, 6, 3, 'bf16_t8')]:       # (8-bit tape: the same tape scales from the first call on)
    geo = synthetic.synthetic_geodesics(H, H, G, S=S, seed=2)
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=depth, net_width=width, mode=mode, device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], geo['J'] if S else None, geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(depth, width).init(1, 21)))
    B = 8
    tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = torch.rand((B, max(S, 1), geom.R), device=dev) - 0.4
    ref = ref_img = None
    bad = 0
    for it in range(40):
        if it % 2:                                   # odd iterations: HBM contention from a second stream
            with torch.cuda.stream(side):
                for _ in range(6):
                    big2.copy_(big)
        img = eng.render_train(geom, tM0).clone()
        g = eng.render_bwd_tape(geom, tM0, dimg).clone()
        torch.cuda.synchronize()
        if ref is None:
            ref, ref_img = g, img
        else:
            bad += int(not torch.equal(g, ref)) + int(not torch.equal(img, ref_img)) * 0   # (the image uses float atomics per ray)
            if not torch.equal(g, ref):
                print('   iteration %d differs: max %.3e' % (it, float((g - ref).abs().max() / ref.abs().max())))
    print('H=%d G=%d %dx%d S=%d %s: %d of 39 repetitions differ' % (H, G, depth, width, S, mode, bad), flush=True)
