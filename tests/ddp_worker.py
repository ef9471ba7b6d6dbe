"""Worker of tests/test_gpu_ddp.py (one rank; started by torch.distributed.run).  All ranks use cuda:0 and the gloo
backend -- a 1-GPU box exercises the real multi-process step: frame sharding, ONE ray set for all ranks per step, the
single gradient all-reduce, identical Adam.  Rank 0 then replays the same steps in one process (whole batch, gradient
divided by the world size: the reference's pmean of per-device sums, network.py:617-621) and writes the verdict."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    # BHNERF_DDP_BACKEND=nccl (world size 1 on a 1-GPU box): the process group of the shipped multi-GPU design -- RCCL --
    # with its device-side all-reduce and the stream semantics of the async work handle (TrainState.exchange_overlapped)
    backend = os.environ.get('BHNERF_DDP_BACKEND', 'gloo')
    if backend == 'nccl':
        assert world == 1, 'every rank of this worker uses cuda:0; RCCL needs one device per rank'
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from bhnerf_amd import constants, engine, network, optimization, synthetic, units
    H = W = 16; G = 32; nt = 8; nsteps = 3; batch = 4; lr = 1e-3
    geos = [synthetic.synthetic_geodesics(H, W, G, seed=s) for s in (1, 2, 3)]          # three "sub-pixel" ray sets
    rts = [network.raytracing_args(dict(x=g['coords'][0], y=g['coords'][1], z=g['coords'][2], dtau=g['dtau'], Sigma=g['Sigma'],
                                        t=g['t_geos'], g=g['g']), g['Omega'], g['t_injection'], 0.0 * units.hr) for g in geos]
    t = np.linspace(0, 1, nt)
    GM = constants.GM_c3('hr')
    target = synthetic.hotspot_movie(geos[0], t, GM)
    overlap = os.environ.get('BHNERF_DDP_OVERLAP') == '1'       # opt-in stale-gradient overlap of the all-reduce (TrainState)
    hp = {'num_iters': nsteps, 'lr_init': lr, 'lr_final': 1e-4, 'seed': 1, 'overlap_allreduce': overlap}
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_width=64, mode='f32', device=dev)
    step = optimization.TrainStep.image(t * units.hr, target, dtype='full')
    opt = optimization.Optimizer(hp, pred, rts)
    p0 = opt.state.flat.clone()
    opt.run(batch, step, rts)
    losses = torch.as_tensor(opt.loss).cpu().numpy()
    flat = opt.state.flat if backend == 'nccl' else opt.state.flat.cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    flat, gathered = flat.cpu(), [g.cpu() for g in gathered]
    out = None
    if rank == 0:
        identical = all(torch.equal(g, gathered[0]) for g in gathered)
        # single-process replay: same frame batches and ray-set choices (same shared generators), whole batch on one
        # device, gradient / world
        pred1 = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_width=64, mode='f32', device=dev)
        st = pred1.init_state(pred1.init_params(rts, seed=1), num_iters=nsteps, lr_init=lr, lr_final=1e-4)
        frames_rng = optimization.TemporalBatchedArgs(t * units.hr, [target]).sample
        rays_rng = optimization._shared_rng(1)
        eng = pred1.engine()
        picks, pending = [], None
        for it in range(nsteps):
            idx = frames_rng(batch)
            k = int(rays_rng.integers(len(rts)))
            picks.append(k)
            rt, geo = rts[k], geos[k]
            geom = pred1.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
            tM0 = engine.frame_offsets(t[idx], 0.0, geo['t_injection'], GM, dev)
            eng.pack(st.flat)
            img = eng.render(geom, tM0)
            tg = torch.as_tensor(target[idx], device=dev).reshape(batch, 1, -1)
            loss, dimg = engine.chi2_image(img, tg, torch.ones_like(tg), torch.zeros_like(tg), 1.0, 'full')
            grad = eng.render_bwd(geom, tM0, dimg)
            if not overlap:
                st.apply_gradients(grad, grad_scale=1.0 / world)
            else:                               # the gradient of step k is applied after the backward of step k+1
                if pending is not None:
                    st.apply_gradients(pending, grad_scale=1.0 / world)
                pending = grad.clone()
        if pending is not None:
            st.apply_gradients(pending, grad_scale=1.0 / world)
        diff = (st.flat.cpu() - flat).abs().numpy()
        moved = float((flat - p0.cpu()).abs().max())
        out = dict(identical=bool(identical), moved=moved, max_diff=float(diff.max()),
                   frac_off=float((diff > 1e-3 * moved).mean()), picks=picks, loss_vector=[float(v) for v in losses],
                   last_loss_single=float(loss.item()), world=world, backend=dist.get_backend(),
                   bitwise_equal_single=bool(torch.equal(st.flat.cpu(), flat)))
        with open(os.environ['BHNERF_DDP_OUT'], 'w') as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
