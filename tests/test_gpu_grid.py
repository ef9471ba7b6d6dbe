"""GPU parity of GRID_Predictor (network.py:254-370): the learnable voxel-grid predictor through the C ABI
(bhn_grid_predict_fwd / bhn_grid_render_fwd / bhn_grid_render_bwd) against golden vectors produced by the reference's
own GRID_Predictor.__call__ and against the float64 oracle's autograd gradient.

Tolerances: emission / images 1e-5 of the maximum (f32, fast exp); gradient 1e-4 of its maximum (f32 atomics)."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from oracle import oracle_torch as ot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def setup(g, dev):
    from bhnerf_amd import network
    sc, rmin, rmax, zw, res = g['hparams']
    pred = network.GRID_Predictor(float(sc), float(rmin), float(rmax), float(zw), int(res), device=dev)
    f = lambda k: np.ascontiguousarray(g[k].astype(np.float32))
    rt = dict(coords=f('coords'), Omega=f('Omega'), g=f('g'), dtau=f('dtau'), Sigma=f('Sigma'), t_geos=f('t_geos'))
    return pred, rt


def test_grid_predictor_forward_golden(dev, golden):
    from bhnerf_amd import network, units
    g = golden('g9_grid')
    pred, rt = setup(g, dev)
    params = {'grid': g['grid'].astype(np.float32)}
    e = pred.apply({'params': params}, g['t_frames'], units.hr, rt['coords'], rt['Omega'], 0.0, rt['t_geos'], float(g['t_injection']))
    assert tuple(e.shape) == g['emission'].shape
    assert np.abs(e.cpu().numpy() - g['emission']).max() <= 1e-5 * g['emission'].max()
    images = network.image_plane_prediction(params, pred.apply, g['t_frames'], rt['coords'], rt['Omega'], 1.0, rt['g'], rt['dtau'],
                                            rt['Sigma'], 0.0, rt['t_geos'], float(g['t_injection']), units.hr)
    assert np.abs(images.cpu().numpy() - g['images']).max() <= 1e-5 * g['images'].max()
    # fused render kernel
    eng = pred.engine()
    geom = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
    eng.pack(eng.flatten(params))
    tM0, _ = network._frame_offsets(g['t_frames'], units.hr, 0.0, float(g['t_injection']), dev)
    img2 = eng.render(geom, tM0)[:, 0].reshape(g['images'].shape)
    assert np.abs(img2.cpu().numpy() - g['images']).max() <= 1e-5 * g['images'].max()
    # sample_3d_grid on the grid's own voxel centres returns sigmoid(grid - 10) inside the domain
    sc = float(g['hparams'][0])
    res = int(g['hparams'][4])
    wide = network.GRID_Predictor(sc, 0.0, np.inf, np.inf, res, device=dev)
    vol = network.sample_3d_grid(wide.apply, params, fov=2 * sc, resolution=res)
    assert np.abs(vol - 1.0 / (1.0 + np.exp(10.0 - g['grid']))).max() < 2e-5


def test_grid_predictor_gradient_vs_oracle_and_reference_fd(dev, golden):
    from bhnerf_amd import engine, network, units
    g = golden('g9_grid')
    pred, rt = setup(g, dev)
    eng = pred.engine()
    geom = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
    flat = eng.flatten({'grid': g['grid'].astype(np.float32)})
    eng.pack(flat)
    tM0, _ = network._frame_offsets(g['t_frames'], units.hr, 0.0, float(g['t_injection']), dev)
    images = eng.render(geom, tM0)
    B, R = images.shape[0], images.shape[2]
    t32 = lambda v: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32), device=dev).reshape(B, 1, R)
    loss, dimg = engine.chi2_image(images, t32(g['target']), t32(g['sigma']), torch.zeros_like(images), 1.0, 'full')
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * float(g['loss'])
    grad = eng.render_bwd(geom, tM0, dimg).reshape(g['grid'].shape).cpu().numpy().astype(np.float64)
    t = lambda v: torch.tensor(np.asarray(v, dtype=np.float64))
    geom_t = dict(coords=t(g['coords']), Omega=t(g['Omega']), t_geos=t(g['t_geos']), g=t(g['g']), dtau=t(g['dtau']), Sigma=t(g['Sigma']),
                  t_start_obs=0.0, t_injection=float(g['t_injection']))
    sc, rmin, rmax, zw, res = g['hparams']
    hp = dict(GM_c3=onp.GM_C3_SGRA_HR, scale=float(sc), rmin=float(rmin), rmax=float(rmax), z_width=float(zw))
    _, _, gref = ot.grid_loss_and_grad(g['grid'], t(g['t_frames']), geom_t, hp, t(g['target']), t(g['sigma']))
    gref = gref.numpy()
    assert np.abs(gref).max() > 0 and np.abs(grad - gref).max() <= 1e-4 * np.abs(gref).max()
    for (i, j, k), fd in zip(g['fd_idx'], g['fd_val']):                       # the reference's own finite differences
        assert abs(grad[i, j, k] - fd) <= 1e-4 * np.abs(gref).max() + 1e-3 * abs(fd)


def test_grid_predictor_training_checkpoint_and_lightcurve(dev, golden, tmp_path):
    """The driver flow with the grid predictor: Optimizer.run with a light-curve loss, flax-format checkpoint, resume."""
    from bhnerf_amd import checkpoints, network, optimization, units
    g = golden('g9_grid')
    pred, rt0 = setup(g, dev)
    rt = network.raytracing_args(dict(x=rt0['coords'][0], y=rt0['coords'][1], z=rt0['coords'][2], dtau=rt0['dtau'], Sigma=rt0['Sigma'],
                                      t=rt0['t_geos'], g=rt0['g']), rt0['Omega'], float(g['t_injection']), 0.0 * units.hr, J=1.0)
    lc = g['images'].sum(axis=(-1, -2))
    step = optimization.TrainStep.image(g['t_frames'] * units.hr, lc, sigma=float(lc.mean()) * 0.1, dtype='lc')
    ckpt = str(tmp_path / 'grid')
    opt = optimization.Optimizer({'num_iters': 40, 'lr_init': 5e-2, 'lr_final': 1e-2}, pred, rt, save_period=20, checkpoint_dir=ckpt)
    # start from a perturbed copy of the golden grid (the reference's own init, -10 everywhere, has a vanishing gradient)
    opt.state.flat.copy_(pred.engine().flatten({'grid': (g['grid'] - 1.0).astype(np.float32)}))
    first = optimization.total_movie_loss(3, opt.state, step, rt)
    opt.run(3, step, rt)
    last = optimization.total_movie_loss(3, opt.state, step, rt)
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    sd = checkpoints.restore_checkpoint(ckpt, None)
    res = int(g['hparams'][4])
    assert int(sd['step']) == 40 and sd['params']['grid'].shape == (res, res, res) and sd['opt_state']['0']['mu']['grid'].shape == (res,) * 3
    pred2 = network.GRID_Predictor.from_yml(ckpt, device=dev)
    assert pred2.grid_res == res and pred2.scale == pred.scale
    st = pred2.init_state(pred2.init_params(rt), checkpoint_dir=ckpt)
    assert st.step == 40 and torch.equal(st.flat, opt.state.flat) and torch.equal(st.m, opt.state.m)
    assert float(pred2.init_params(rt).flat.max()) == -10.0
