import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from bhnerf_amd import engine, network
dev = torch.device('cuda:0')
def case(width, depth, S, mode, H=9, Wd=7, G=50, B=3, reps=6):
    rng = np.random.default_rng(1)
    P = H * Wd * G
    coords = rng.uniform(-6, 6, (3, H, Wd, G)).astype(np.float32)
    geo = engine.RayGeometry(coords, rng.uniform(0.01, 0.1, (H, Wd, G)).astype(np.float32), rng.uniform(0.6, 1.4, (H, Wd, G)).astype(np.float32),
                             np.full((H, Wd, G), 0.1, np.float32), np.ones((H, Wd, G), np.float32), np.full((H, Wd, G), -990.0, np.float32),
                             (rng.uniform(-1, 1, (S, H, Wd, G)).astype(np.float32) if S else None), 0.0, 100.0, 100.0, dev)
    pred = network.NeRF_Predictor(8.0, 0.0, 100.0, 100.0, net_depth=depth, net_width=width, mode=mode, device=dev)
    eng = pred.engine()
    flat = eng.flatten(network.MLP(depth, width).init(3, 21))
    eng.pack(flat)
    tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, -1000.0, 0.00568, dev)
    dimg = torch.randn((B, max(S, 1), geo.R), device=dev)
    outs = [eng.render_bwd(geo, tM0, dimg).clone() for _ in range(reps)]
    imgs = [eng.render(geo, tM0).clone() for _ in range(3)]
    ref = outs[0]
    diffs = [float((o - ref).abs().max() / ref.abs().max()) for o in outs[1:]]
    lay = []
    for o in outs[1:]:
        d = (o - ref).abs()
        lay.append(['%.0e' % float(d[eng.kernel_off[i]:eng.bias_off[i] + eng.out_dim[i]].max() / ref.abs().max()) for i in range(depth + 1)])
    print(width, depth, S, mode, 'G', G, 'run-to-run grad diffs', ['%.1e' % d for d in diffs], 'img', float((imgs[1] - imgs[0]).abs().max()))
    print('   per-layer', lay[:3])
for mode in ('f32', 'bf16'):
    case(128, 4, 3, mode); case(128, 4, 0, mode); case(256, 4, 0, mode); case(64, 8, 2, mode); case(128, 4, 3, mode, G=64, H=16, Wd=16, B=4)
