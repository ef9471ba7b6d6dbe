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
