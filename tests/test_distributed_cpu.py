"""world_size-2 gloo tests (CPU) of the data-parallel path: frame sharding and the single gradient
all-reduce reproduce the reference's mean-of-per-device-sums semantics (network.py:477/620)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from bhnerf_amd import network, optimization
        from oracle import oracle_torch as ot
        from conftest import golden_tree
        import test_oracle_golden as tg
        g = dict(np.load(os.path.join(root, 'tests', 'golden', 'g5_predict_e.npz')))      # 4 frames, 4x128
        assert optimization.device_count() == world and network._world() == (rank, world)
        idx = np.array([2, 0, 3, 1])                                  # the batch every rank draws
        mine = optimization.shard(idx)                                 # contiguous slice (optimization.py:360-362)
        assert np.array_equal(mine, idx[rank * 2:(rank + 1) * 2])
        with pytest.raises(ValueError):
            optimization.shard(np.arange(3))
        # per-device SUM of chi^2 over this rank's frames and its gradient, with the oracle
        tr, t = tg._torch_trainer(g)
        shape = (4,) + g['coords'].shape[1:3]
        tgt = {k: t(g[k + '_full']).reshape(shape) for k in ('target', 'sigma', 'offset')}
        loss, _, grads = tr.loss_and_grad(t(g['t_frames'][mine]), tgt['target'][mine], tgt['sigma'][mine], tgt['offset'][mine], 1.0, 'full')
        n = sum(p.numel() for p in grads)
        buf = torch.zeros(n + world, dtype=torch.float64)
        buf[:n] = torch.cat([p.reshape(-1) for p in grads])
        loss_vec = network.dp_allreduce(buf, n, loss.reshape(1), rank, world)
        out.put((rank, loss_vec.numpy().copy(), (buf[:n] / world).numpy().copy(), float(loss)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_exchange_matches_mean_of_sums():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # both ranks hold identical averaged gradients and the full loss vector
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][1], res[1][1])
    assert np.allclose(res[0][1], [res[0][3], res[1][3]])
    # reference semantics: grad = (1/ndev) * sum over ALL frames of the batch of d chi^2_frame
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_oracle_golden as tg
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g5_predict_e.npz')))
    tr, t = tg._torch_trainer(g)
    shape = (4,) + g['coords'].shape[1:3]
    tgt = {k: t(g[k + '_full']).reshape(shape) for k in ('target', 'sigma', 'offset')}
    loss, _, grads = tr.loss_and_grad(t(g['t_frames']), tgt['target'], tgt['sigma'], tgt['offset'], 1.0, 'full')
    ref = torch.cat([p.reshape(-1) for p in grads]).numpy() / world
    assert np.allclose(res[0][2], ref, rtol=1e-10, atol=1e-18)
    assert np.isclose(res[0][1].sum(), float(loss), rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# The step driver itself (network._step_image: pack -> render -> chi^2 -> backward -> ONE all-reduce -> Adam with
# grad / world) under gloo, with the device engine replaced by an oracle-backed stand-in (no GPU here).  What is under
# test is the sequencing and the data-parallel arithmetic of bhnerf_amd.network / optimization, not the kernels.
# ---------------------------------------------------------------------------------------------------------------
class _StubEngine:
    def __init__(self, trainer, t_start_obs, t_injection, GM_c3):
        self.tr, self.device = trainer, torch.device('cpu')
        self.t0, self.tinj, self.GM = float(t_start_obs), float(t_injection), float(GM_c3)
        self.nl = len(trainer.k)
        self.nparams = sum(p.numel() for p in trainer.k + trainer.b)
        self.calls = []

    def _params_flax_order(self):
        return [p for i in range(self.nl) for p in (self.tr.k[i], self.tr.b[i])]

    def flatten(self):
        return torch.cat([p.detach().reshape(-1) for p in self._params_flax_order()]).float()

    def pack(self, flat):
        self.calls.append('pack')
        off = 0
        with torch.no_grad():
            for p in self._params_flax_order():
                p.copy_(flat[off:off + p.numel()].reshape(p.shape).double())
                off += p.numel()

    def fits_tape(self, B, P):
        return True

    def render_train(self, geom, tM0, out=None):
        self.calls.append('render')
        t_frames = (tM0.double() + self.tinj) * self.GM + self.t0
        self._images = self.tr.forward(t_frames)
        return self._images.detach().float().reshape(tM0.numel(), 1, -1)

    render = render_train

    def render_bwd_tape(self, geom, tM0, dimg, out=None):
        self.calls.append('bwd')
        params = self._params_flax_order()
        grads = torch.autograd.grad(self._images, params, grad_outputs=dimg.reshape(self._images.shape).double())
        out.copy_(torch.cat([g.reshape(-1) for g in grads]).float())
        return out


def _frames_for(world):
    rep = max(1, world // 4)
    return 4 * rep, rep


def _batch_indices(it, nf):
    base = np.array([[2, 0, 3, 1], [0, 1, 2, 3], [3, 2, 1, 0]][it])
    return np.concatenate([base + 4 * r for r in range(nf // 4)]) if nf > 4 else base


def _step_worker(rank, world, port, out, overlap=False):
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from bhnerf_amd import constants, network, optimization, units
    from oracle import oracle_np as onp
    import test_oracle_golden as tg
    g = dict(np.load(os.path.join(root, 'tests', 'golden', 'g5_predict_e.npz')))      # 4 frames, 4x128
    tr, t = tg._torch_trainer(g)
    assert abs(constants.GM_c3('hr') - onp.GM_C3_SGRA_HR) < 1e-12 * onp.GM_C3_SGRA_HR
    eng = _StubEngine(tr, g['t_start_obs'], g['t_injection'], onp.GM_C3_SGRA_HR)
    H, W = g['coords'].shape[1:3]
    pred = types.SimpleNamespace(engine=lambda: eng, geometry=lambda *a, **k: types.SimpleNamespace(P_eff=H * W, S=0, Sx=1, R=H * W, spatial=(H, W)))

    def chi2_image(images, target, sigma, offset, scale, dtype, want_grad=True):       # engine.chi2_image ('full')
        d = (images - target - offset) / sigma
        return (scale * (d * d).sum()).reshape(1), (2.0 * scale * d / sigma if want_grad else None)

    def adam_step(params, grads, m, v, tcount, lr, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):   # bhn_adam_step
        eng.calls.append('adam(grad_scale=%g)' % grad_scale)
        gs = grads * grad_scale
        m.mul_(b1).add_(gs, alpha=1 - b1)
        v.mul_(b2).addcmul_(gs, gs, value=1 - b2)
        params.sub_(lr * (m / (1 - b1 ** tcount)) / ((v / (1 - b2 ** tcount)).sqrt() + eps))

    network.engine.chi2_image, network.engine.adam_step = chi2_image, adam_step
    # the state is built BEFORE the process group exists (ADVICE: the gradient buffer must follow the world size)
    state = network.TrainState(None, eng.flatten(), pred, 3, 1e-3, 1e-4)
    state.overlap_allreduce = overlap
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        shape = (4, H, W)
        nf, rep = _frames_for(world)            # world 8: the fixture's four frames twice (one frame per rank)
        arrs = [np.concatenate([g[k + '_full'].reshape(shape).astype(np.float32)] * rep) for k in ('target', 'sigma', 'offset')]
        batched = optimization.TemporalBatchedArgs(np.concatenate([g['t_frames']] * rep) * units.hr, arrs)
        losses = []
        for it in range(3):
            idx = _batch_indices(it, nf)
            tgt, sig, off, tf = batched[idx]                                          # this rank's contiguous slice
            assert len(tf) == nf // world
            loss, state, images = network.gradient_step_image(
                state, units.hr, 'full', tgt, sig, off, tf, g['coords'], g['Omega'], 1.0, g['g'], g['dtau'], g['Sigma'],
                float(g['t_start_obs']), g['t_geos'], float(g['t_injection']), 1.0)
            losses.append(loss.numpy().copy())
            assert images.shape == (1, nf // world, H, W)
        adam = 'adam(grad_scale=%g)' % (1.0 / world)
        if overlap:
            # the all-reduce of step k completes inside step k+1 (after its backward); the last one at finish_allreduce()
            assert eng.calls == ['pack', 'render', 'bwd'] + ['pack', 'render', 'bwd', adam] * 2, eng.calls
            assert state.step == 2 and np.all(np.isfinite(losses[0])) and np.all(losses[0] == losses[0][rank])     # first step: own loss in every slot
            state.finish_allreduce()
            assert eng.calls[-1] == adam and state._pending is None
            losses = losses[1:]                  # steps 2 and 3 return the loss vectors of the completed steps 1 and 2
        else:
            assert eng.calls == ['pack', 'render', 'bwd', adam] * 3, eng.calls
        assert state.grad.numel() == eng.nparams + world and state.step == 3
        out.put((rank, state.flat.numpy().copy(), np.array(losses)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 4, 8])
def test_two_rank_training_steps_match_single_process_reference(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1])                        # identical parameters on every rank, bitwise
        assert np.array_equal(res[0][2], r[2])                        # every rank holds every rank's loss
    assert res[0][2].shape == (3, world)
    # reference: one process, all four frames, gradient / world (pmean of per-device sums), float64
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_oracle_golden as tg
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g5_predict_e.npz')))
    tr, t = tg._torch_trainer(g)
    tr.num_iters, tr.lr_init, tr.lr_final = 3, 1e-3, 1e-4
    p0 = torch.cat([p.detach().reshape(-1) for i in range(len(tr.k)) for p in (tr.k[i], tr.b[i])]).numpy().copy()
    shape = (4,) + g['coords'].shape[1:3]
    nf, rep = _frames_for(world)
    tgt = {k: torch.cat([t(g[k + '_full']).reshape(shape)] * rep) for k in ('target', 'sigma', 'offset')}
    t_all = np.concatenate([g['t_frames']] * rep)
    ref_losses = []
    for it in range(3):
        idx = _batch_indices(it, nf)
        loss, _ = tr.step(t(t_all[idx]), tgt['target'][idx], tgt['sigma'][idx], tgt['offset'][idx], 1.0, 'full', grad_div=world)
        ref_losses.append(float(loss))
    ref = torch.cat([p.detach().reshape(-1) for i in range(len(tr.k)) for p in (tr.k[i], tr.b[i])]).numpy()
    moved = np.abs(ref - p0).max()
    assert moved > 1e-4
    assert np.abs(res[0][1] - ref).max() < 2e-3 * moved               # float32 Adam state vs the float64 reference
    assert np.allclose(res[0][2].sum(axis=1), ref_losses, rtol=2e-5)  # sum of the per-device sums = chi^2 of the batch


@pytest.mark.timeout(300)
def test_two_rank_overlapped_allreduce_is_the_one_step_stale_update():
    """overlap_allreduce (opt-in): the gradient of step k is applied after the backward of step k+1, identically on
    both ranks, and equals a single-process run that applies the (all-frames, / world) gradient one step late."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, out, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])                       # identical parameters on both ranks, bitwise
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_oracle_golden as tg
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g5_predict_e.npz')))
    tr, t = tg._torch_trainer(g)
    tr.num_iters, tr.lr_init, tr.lr_final = 3, 1e-3, 1e-4
    p0 = torch.cat([p.detach().reshape(-1) for i in range(len(tr.k)) for p in (tr.k[i], tr.b[i])]).numpy().copy()
    shape = (4,) + g['coords'].shape[1:3]
    tgt = {k: t(g[k + '_full']).reshape(shape) for k in ('target', 'sigma', 'offset')}
    pending, ref_losses = None, []
    for it in range(3):
        idx = [[2, 0, 3, 1], [0, 1, 2, 3], [3, 2, 1, 0]][it]
        loss, _, grads = tr.loss_and_grad(t(g['t_frames'][idx]), tgt['target'][idx], tgt['sigma'][idx], tgt['offset'][idx], 1.0, 'full')
        ref_losses.append(float(loss))
        if pending is not None:
            tr.apply(pending, grad_div=world)
        pending = [gr.detach().clone() for gr in grads]
    tr.apply(pending, grad_div=world)
    ref = torch.cat([p.detach().reshape(-1) for i in range(len(tr.k)) for p in (tr.k[i], tr.b[i])]).numpy()
    moved = np.abs(ref - p0).max()
    assert moved > 1e-4
    assert np.abs(res[0][1] - ref).max() < 2e-3 * moved
    # the loss vectors returned by steps 2 and 3 are those of the completed steps 1 and 2
    assert res[0][2].shape == (2, 2) and np.allclose(res[0][2].sum(axis=1), ref_losses[:2], rtol=2e-5)
