"""CPU tests: the oracle (numpy + torch restatements) against golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  Tolerances: float64, 1e-12 relative."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from oracle import oracle_torch as ot
from conftest import golden_tree

PRED = ['a', 'b', 'c', 'd', 'e', 'f']


def close(a, b, rtol=1e-12, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_constants(golden):
    g = golden('g0_constants')
    assert onp.GM_C3_SGRA_HR == float(g['GM_c3_hr'])
    assert float(g['isco0']) == pytest.approx(6.0, abs=1e-12)


def test_warp(golden):
    g = golden('g1_warp')
    a = (g['coords'], g['Omega'], g['t_frames'])
    close(onp.velocity_warp_coords(*a, 0.0, g['t_geos'], float(g['t_injection']), GM_c3=onp.GM_C3_SGRA_HR),
          g['out_units'], atol=1e-12)
    close(onp.velocity_warp_coords(*a, 0.1, g['t_geos'], float(g['t_injection'])), g['out_nounits'], atol=1e-12)
    close(onp.velocity_warp_coords(g['coords'], 0.05, g['t_frames'], 0.0, g['t_geos'], float(g['t_injection']),
                                   GM_c3=onp.GM_C3_SGRA_HR), g['out_scalar_omega'], atol=1e-12)
    close(onp.velocity_warp_coords(g['coords'], g['Omega'], 0.4, 0.0, 0.0, 0.0), g['out_scalar_t'], atol=1e-12)
    close(onp.rotation_matrix([0, 0, 1], g['rot_angles']), g['rot'], atol=1e-15)
    assert 0.05 < float(g['nan_fraction']) < 0.5        # the NaN branch is exercised
    # torch restatement (closed-form z rotation) agrees with the Euler-Rodrigues form
    t = lambda x: torch.tensor(x, dtype=torch.float64)
    w = ot.warp(t(g['coords']), t(g['Omega']), t(g['t_frames']), 0.0, t(g['t_geos']), float(g['t_injection']),
                onp.GM_C3_SGRA_HR).numpy()
    close(w, g['out_units'], atol=1e-11)


def test_posenc(golden):
    g = golden('g2_posenc')
    for d in (0, 1, 3, 5):
        close(onp.posenc(g['x'], d), g['deg%d' % d], atol=1e-15)
    close(ot.posenc(torch.tensor(g['x']), 3).numpy(), g['deg3'], atol=1e-13)
    close(g['deg3'][0, :6], [.1, -.2, .3, 0.09983341664682815, -0.19866933079506122, 0.29552020666133955], atol=1e-13)


def test_fill(golden):
    g = golden('g3_fill')
    out = onp.fill_unsupervised_emission(g['emission'], g['coords'], float(g['rmin']), float(g['rmax']),
                                         float(g['z_width']))
    assert np.array_equal(out, g['out'])
    assert (g['hits'] > 0).all()                        # each of the three masks is hit


def test_radiative_transfer(golden):
    g = golden('g4_rt')
    close(onp.radiative_trasfer(g['emission'], g['g'], g['dtau'], g['Sigma']), g['out_arrays'])
    close(onp.radiative_trasfer(g['emission'][0, 0], 1.3, 1.0, 0.5), g['out_scalars'])
    close(onp.radiative_trasfer(g['emission'][1, 2], g['g'], g['dtau'], g['Sigma']), g['out_3d'])


def _np_forward(g):
    hp = g['hparams']
    e = onp.predictor_apply(golden_tree(g), g['t_frames'], g['coords'], g['Omega'], float(g['t_start_obs']),
                            g['t_geos'], float(g['t_injection']), scale=hp[0], rmin=hp[1], rmax=hp[2],
                            z_width=hp[3], posenc_deg=int(hp[4]), net_depth=int(hp[5]))
    J = g['J'] if g['J'].ndim else float(g['J'])
    return e, onp.image_plane_prediction(e, J, g['g'], g['dtau'], g['Sigma'])


@pytest.mark.parametrize('tag', PRED)
def test_predictor_and_render_numpy(golden, tag):
    g = golden('g5_predict_' + tag)
    e, images = _np_forward(g)
    close(e, g['emission'], rtol=1e-11)
    assert images.shape == g['images'].shape            # incl. the b=1 squeeze quirk (tag d)
    close(images, g['images'], rtol=1e-11)
    for dt in ('full', 'lc'):
        l = onp.loss_image(images, g['target_' + dt], g['sigma_' + dt], g['offset_' + dt], g['hparams'][7], dt)
        close(l, g['loss_' + dt], rtol=1e-11)
    assert (g['emission'] == 0).mean() > 0.1 and (g['emission'] > 0).mean() > 0.05


def _torch_trainer(g, dtype=torch.float64):
    hp = g['hparams']
    ks, bs = ot.tree_to_lists(golden_tree(g), dtype)
    t = lambda x: torch.tensor(x, dtype=dtype)
    geom = dict(coords=t(g['coords']), Omega=t(g['Omega']), t_geos=t(g['t_geos']), g=t(g['g']), dtau=t(g['dtau']),
                Sigma=t(g['Sigma']), J=(t(g['J']) if g['J'].ndim else None), t_start_obs=float(g['t_start_obs']),
                t_injection=float(g['t_injection']))
    hpd = dict(GM_c3=onp.GM_C3_SGRA_HR, scale=hp[0], rmin=hp[1], rmax=hp[2], z_width=hp[3], posenc_deg=int(hp[4]),
               net_depth=int(hp[5]))
    return ot.CpuTrainer(ks, bs, geom, hpd), t


@pytest.mark.parametrize('tag', PRED)
def test_predictor_and_gradient_torch(golden, tag):
    g = golden('g5_predict_' + tag)
    tr, t = _torch_trainer(g)
    S = g['J'].shape[0] if g['J'].ndim else None
    b = len(g['t_frames'])
    shape = (b, S) + g['coords'].shape[1:3] if S else (b,) + g['coords'].shape[1:3]
    tgt = {k: t(g[k + '_full']).reshape(shape) for k in ('target', 'sigma', 'offset')}
    loss, images, grads = tr.loss_and_grad(t(g['t_frames']), tgt['target'], tgt['sigma'], tgt['offset'],
                                           float(g['hparams'][7]), 'full')
    close(images.numpy().reshape(g['images'].shape), g['images'], rtol=1e-10)
    close(loss.item(), g['loss_full'], rtol=1e-10)
    nl = len(tr.k)
    for (li, i, j), fd in zip(g['fd_idx'], g['fd_val']):
        ana = grads[li][i, j].item() if i >= 0 else grads[nl + li][j].item()
        scale = max(abs(fd), float(np.abs(g['fd_val']).max()) * 1e-3)
        assert abs(ana - fd) <= 2e-5 * scale, (li, i, j, ana, fd)
    assert np.abs(g['fd_val']).max() > 0


def test_loss_eht(golden):
    g = golden('g7_eht')
    s = float(g['scale'])
    close(onp.loss_eht(g['images'], g['target_vis'], g['sigma'], g['A'], s, 'vis'), g['loss_vis'])
    close(onp.loss_eht(g['images'], g['target_amp'], g['sigma'], g['A'], s, 'amp'), g['loss_amp'])
    close(onp.loss_eht(g['images'], g['target_cphase'], g['sigma'], g['A3'], s, 'cphase'), g['loss_cphase'])
    with pytest.raises(AttributeError):
        onp.loss_eht(g['images'], g['target_vis'], g['sigma'], g['A'], s, 'nope')


def test_adam_matches_torch_optim():
    """optax.adam == torch.optim.Adam update form (eps added to sqrt(v_hat)); linear decay schedule."""
    rng = np.random.default_rng(0)
    p0 = rng.normal(size=(5, 3)); p = p0.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    tp = torch.tensor(p0, requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1.0, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 6):
        gnp = rng.normal(size=p.shape)
        lr = onp.linear_lr(t - 1, 1e-2, 1e-4, 4)
        p, m, v = onp.adam_step(p, gnp, m, v, t, lr)
        for grp in opt.param_groups:
            grp['lr'] = lr
        tp.grad = torch.tensor(gnp)
        opt.step()
        close(tp.detach().numpy(), p, rtol=1e-12)
    assert onp.linear_lr(0, 1e-2, 1e-4, 4) == 1e-2 and onp.linear_lr(9, 1e-2, 1e-4, 4) == 1e-4


def test_mlp_dims():
    assert onp.mlp_layer_dims(4, 256, 21) == [(21, 256), (256, 256), (256, 256), (277, 256), (256, 1)]
    assert [d[0] for d in onp.mlp_layer_dims(8, 256, 21)] == [21, 256, 256, 256, 256, 277, 256, 256, 256]
    n = sum(a * b + b for a, b in onp.mlp_layer_dims(4, 256, 21))
    assert n == 208641 and sum(a * b + b for a, b in onp.mlp_layer_dims(4, 128, 21)) == 55169


def test_grid_predictor_golden(golden):
    """GRID_Predictor.__call__ of the reference (network.py:306-353) vs the oracle: emission, images, chi^2 and its
    finite-difference gradient w.r.t. 40 voxels (float64)."""
    import torch
    from oracle import oracle_torch as ot
    g = golden('g9_grid')
    sc, rmin, rmax, zw, res = g['hparams']
    e = onp.grid_predictor_apply(g['grid'], g['t_frames'], g['coords'], g['Omega'], 0.0, g['t_geos'], float(g['t_injection']),
                                 scale=sc, rmin=rmin, rmax=rmax, z_width=zw)
    assert e.shape == g['emission'].shape and np.abs(e - g['emission']).max() < 1e-12
    assert 0.05 < (e > 0).mean() < 0.5 and np.isclose(e[e > 0].min(), 1.0 / (1.0 + np.exp(10.0)), rtol=1e-6)   # valid points outside the grid
    img = onp.radiative_trasfer(e, g['g'], g['dtau'], g['Sigma'])
    assert np.abs(img - g['images']).max() < 1e-11
    t = lambda v: torch.tensor(np.asarray(v, dtype=np.float64))
    geom = dict(coords=t(g['coords']), Omega=t(g['Omega']), t_geos=t(g['t_geos']), g=t(g['g']), dtau=t(g['dtau']), Sigma=t(g['Sigma']),
                t_start_obs=0.0, t_injection=float(g['t_injection']))
    hp = dict(GM_c3=onp.GM_C3_SGRA_HR, scale=float(sc), rmin=float(rmin), rmax=float(rmax), z_width=float(zw))
    loss, img_t, grad = ot.grid_loss_and_grad(g['grid'], t(g['t_frames']), geom, hp, t(g['target']), t(g['sigma']))
    assert abs(loss.item() - float(g['loss'])) < 1e-9 * float(g['loss'])
    assert int(g['nonzero_fd']) >= 10
    for (i, j, k), fd in zip(g['fd_idx'], g['fd_val']):
        assert abs(grad[i, j, k].item() - fd) <= 1e-6 * max(1.0, abs(fd)), ((i, j, k), grad[i, j, k].item(), fd)


def test_image_plane_dynamics_golden(golden):
    """The oracle's composition of emission.image_plane_dynamics (warp -> trilinear sampling -> x J -> radiative transfer)
    against the reference's own output (fixture g8): the checker of the full-size BASELINE config-1 GPU test."""
    g = golden('g8_dynamics')
    fov = float(g['axis'][-1] - g['axis'][0])
    kw = dict(t_geos=g['t_geos'], g=1.0, dtau=g['dtau'], Sigma=g['Sigma'])
    img = onp.image_plane_dynamics(g['volume'], fov, g['coords'], g['Omega'], g['t_frames'], float(g['t_injection']), **kw)
    assert img.shape == g['images'].shape and np.abs(img - g['images']).max() < 1e-12 * np.abs(g['images']).max()
    imgJ = onp.image_plane_dynamics(g['volume'], fov, g['coords'], g['Omega'], g['t_frames'], float(g['t_injection']), J=g['J'], **kw)
    assert imgJ.shape == g['images_J'].shape and np.abs(imgJ - g['images_J']).max() < 1e-12 * np.abs(g['images_J']).max()
    fast = onp.image_plane_dynamics(g['volume'], fov, g['coords'], g['Omega'], g['t_frames'], float(g['t_injection']),
                                    t_start_obs=0.1, slow_light=False, **kw)
    assert np.abs(fast - g['images_fast']).max() < 1e-12 * np.abs(g['images_fast']).max()

