"""GPU parity of the EHT visibility-domain losses against the golden vectors produced by
the reference's own loss_fn_eht and against torch complex autograd (float64) for the image gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _ref_loss(images, A, target, sigma, scale, dtype):
    img = torch.tensor(images, dtype=torch.float64, requires_grad=True)
    At, tg, sg = torch.tensor(A), torch.tensor(target), torch.tensor(sigma)
    vec = img.reshape(img.shape[0], -1, 1).to(torch.complex128)
    if dtype == 'cphase':
        vis = (At.to(torch.complex128) @ vec[:, None]).squeeze(-1)
        loss = scale * ((1.0 - torch.cos(tg - torch.angle(vis.prod(dim=-2)))) / sg ** 2).sum()
    else:
        vis = (At.to(torch.complex128) @ vec).squeeze(-1)
        loss = scale * (((vis - tg).abs() / sg) ** 2).sum() if dtype == 'vis' else scale * (((vis.abs() - tg) / sg).abs() ** 2).sum()
    loss.backward()
    return loss.item(), img.grad.numpy()


@pytest.mark.parametrize('dtype', ['vis', 'amp', 'cphase'])
def test_chi2_eht_golden_and_gradient(dev, golden, dtype):
    from bhnerf_amd import engine
    g = golden('g7_eht')
    A = g['A3'] if dtype == 'cphase' else g['A']
    target = g['target_' + dtype]
    scale = float(g['scale'])
    images = torch.tensor(g['images'], dtype=torch.float32, device=dev)
    loss, dimg = engine.chi2_eht(images.reshape(3, -1), A, target, g['sigma'], scale, dtype)
    assert abs(loss.item() - float(g['loss_' + dtype])) < 2e-5 * abs(float(g['loss_' + dtype]))     # reference's own value
    ref_loss, ref_grad = _ref_loss(g['images'], A, target, g['sigma'], scale, dtype)
    assert abs(ref_loss - float(g['loss_' + dtype])) < 1e-9 * abs(ref_loss)
    err = np.abs(dimg.cpu().numpy().reshape(ref_grad.shape) - ref_grad).max() / np.abs(ref_grad).max()
    assert err < 2e-5, err
    with pytest.raises(AttributeError):
        engine.chi2_eht(images.reshape(3, -1), A, target, g['sigma'], scale, 'nope')
    with pytest.raises(AttributeError):          # ndim contract of network.py:546/556
        engine.chi2_eht(images.reshape(3, -1), g['A3'] if dtype != 'cphase' else g['A'], target, g['sigma'], scale, dtype)


def test_eht_training_step_and_loss_fn(dev):
    """loss_fn_eht through the fused renderer is differentiable w.r.t. the parameters and
    TrainStep.eht_arrays + gradient_step_eht decrease a visibility chi^2 on a synthetic hotspot."""
    from bhnerf_amd import constants, network, observation, optimization, synthetic, units
    H = W = 16; G = 32; nt = 4
    geo = synthetic.synthetic_geodesics(H, W, G, seed=2)
    t_frames = np.linspace(0, 0.5, nt)
    movie = synthetic.hotspot_movie(geo, t_frames, constants.GM_c3('hr'))
    rng = np.random.default_rng(0)
    uv = rng.normal(scale=2e9, size=(10, 2))
    A1 = observation.dft_matrix(uv, fov=1e-10, npix=H)
    A = np.broadcast_to(A1, (nt,) + A1.shape).copy()
    vis = np.einsum('tkp,tp->tk', A, movie.reshape(nt, -1).astype(np.complex64))
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=64, mode='f32', device=dev)
    rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'],
                                      Sigma=geo['Sigma'], t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'],
                                 0.0 * units.hr)
    sigma = np.full(vis.shape, float(np.abs(vis).mean()) * 0.1, dtype=np.float32)
    # (1) autograd through loss_fn_eht == fused step's gradient
    params = pred.init_params(rt, seed=3)
    flat = params.flat.clone().requires_grad_(True)
    tree = network.ParamTree(); tree.flat = flat
    loss, [images] = network.loss_fn_eht(tree, pred.apply, vis, sigma, A, t_frames, *rt.values(), 1.0, units.hr, 'vis')
    loss.backward()
    state = pred.init_state(params, num_iters=50, lr_init=1e-3, lr_final=1e-4)
    l0, state, imgs = network.gradient_step_eht(state, units.hr, 'vis', vis, sigma, A, t_frames, *rt.values(), 1.0)
    assert abs(l0.item() - loss.item()) < 1e-4 * abs(loss.item())
    g_step = state.grad[:flat.numel()]
    assert float((g_step - flat.grad).abs().max()) < 1e-4 * float(flat.grad.abs().max())
    assert imgs.shape == (1, nt, H, W)
    # (2) optimisation loop through TrainStep.eht_arrays
    step = optimization.TrainStep.eht_arrays(t_frames * units.hr, vis, sigma, A, dtype='vis')
    opt = optimization.Optimizer({'num_iters': 40, 'lr_init': 2e-3, 'lr_final': 2e-4, 'seed': 3}, pred, rt)
    first = optimization.total_movie_loss(nt, opt.state, step, rt)
    opt.run(nt, step, rt)
    last = optimization.total_movie_loss(nt, opt.state, step, rt)
    assert opt.state.step == 40 and last < 0.9 * first, (first, last)
    with pytest.raises(AttributeError):
        optimization.TrainStep.eht_arrays(t_frames * units.hr, vis, sigma, A, dtype='nope')
