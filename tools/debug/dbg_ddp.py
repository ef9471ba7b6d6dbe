"""2 ranks on one GPU (gloo): the DP step must equal a single-process step on the same 2x batch with grad/2."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.cuda.set_device(0); dev = torch.device('cuda:0')
from bhnerf_amd import network, optimization, synthetic, units, constants
H = W = 16; G = 32; nt = 8
geo = synthetic.synthetic_geodesics(H, W, G, seed=1)
rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'], t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr)
t = np.linspace(0, 1, nt)
target = synthetic.hotspot_movie(geo, t, constants.GM_c3('hr'))
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_width=64, mode='f32', device=dev)
step = optimization.TrainStep.image(t * units.hr, target, dtype='full')
opt = optimization.Optimizer({'num_iters': 3, 'lr_init': 1e-3, 'lr_final': 1e-4, 'seed': 1}, pred, rt)
p0 = opt.state.flat.clone()
opt.run(4, step, rt)
losses = torch.as_tensor(opt.loss).cpu().numpy()
flat = opt.state.flat.cpu()
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
if rank == 0:
    print('ranks hold identical parameters:', all(torch.equal(g, gathered[0]) for g in gathered), 'loss vector', losses, 'moved', float((flat - p0.cpu()).abs().max()))
    # single-process reference: same batches (same sampler seed), gradient of the whole batch divided by world
    from bhnerf_amd import engine
    pred1 = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_width=64, mode='f32', device=dev)
    st = pred1.init_state(pred1.init_params(rt, seed=1), num_iters=3, lr_init=1e-3, lr_final=1e-4)
    a = optimization.TemporalBatchedArgs(t * units.hr, [target, np.ones_like(target), np.zeros_like(target)])
    eng = pred1.engine()
    geom = pred1.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
    for it in range(3):
        idx = a.sample(4)
        tM0 = engine.frame_offsets(t[idx], 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
        eng.pack(st.flat)
        img = eng.render(geom, tM0)
        tg = torch.as_tensor(target[idx], device=dev).reshape(4, 1, -1)
        loss, dimg = engine.chi2_image(img, tg, torch.ones_like(tg), torch.zeros_like(tg), 1.0, 'full')
        grad = eng.render_bwd(geom, tM0, dimg)
        st.apply_gradients(grad, grad_scale=1.0 / world)
    print('DP == single-process mean-of-sums: max param diff %.3e (movement %.3e)' % (float((st.flat.cpu() - flat).abs().max()), float((flat - p0.cpu()).abs().max())))
dist.destroy_process_group()
