"""BASELINE config 4 at size (Tutorial4: EHT2017 array, 256x256 image, complex-visibility loss; network.py:486-564,
optimization.py:219-268): the (u, v) tracks of the reference's EHT2017 station file (fixture g11, made by
tests/golden/make_eht2017.py) -> direct-DFT matrices -> `loss_fn_eht` on images rendered by the fused kernel from
256x256 rays x 100 samples, 8 frames, against the NumPy oracle and float64 complex autograd; the training step through
`TrainStep.eht_arrays`; achieved bandwidth of the visibility kernels."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

NPIX, G, NT = 256, 100, 8
FOV_M = 16.0
RAD_PER_M = 5.03e-6 / 3600.0 * np.pi / 180.0          # GM/c^2/D of Sgr A* in radians (5.03 micro-arcseconds)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def setup(dev, golden):
    from bhnerf_amd import constants, network, observation, synthetic, units
    from test_gpu_eht import _ref_loss                                     # noqa: F401  (float64 complex autograd)
    g = golden('g11_eht2017')
    frames = np.arange(0, 64, 64 // NT)                                    # 8 of the 64 frames of the track
    t_hr = g['t_hr'][frames]
    geo = synthetic.synthetic_geodesics(NPIX, NPIX, G, fov_M=FOV_M, inc_deg=60.0, seed=0)
    movie = synthetic.hotspot_movie(geo, t_hr, constants.GM_c3('hr'))     # (8, 256, 256) truth
    movie = movie * (2.0 / movie.sum(axis=(1, 2)).mean())                  # ~2 Jy total flux
    fov = FOV_M * RAD_PER_M
    A = np.stack([observation.dft_matrix(g['uv'][f], fov, NPIX) for f in frames])          # (8, 28, 65536) complex64
    assert A.shape == (NT, 28, NPIX * NPIX)
    rng = np.random.default_rng(4)
    vis_true = np.einsum('tkp,tp->tk', A.astype(np.complex128), movie.reshape(NT, -1).astype(np.complex128))
    sigma = g['sigma'][frames].copy()
    noise = (rng.normal(size=vis_true.shape) + 1j * rng.normal(size=vis_true.shape)) * sigma
    target = (vis_true + noise).astype(np.complex64)
    sigma[~g['up'][frames]] = 1e15                                         # baselines with a station below 10 deg carry no weight
    sigma = sigma.astype(np.float32)
    rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'],
                                      Sigma=geo['Sigma'], t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'],
                                 0.0 * units.hr)
    pred = network.NeRF_Predictor(FOV_M / 2, 2.0, FOV_M / 2, 4.0, net_depth=4, net_width=128, mode='f32', device=dev)
    params = pred.init_params(rt, seed=5)
    with torch.no_grad():                              # a non-trivial image: emission sigmoid(out - 10) needs a large bias
        tree = pred.engine().unflatten(params.flat)
        tree['MLP_0']['Dense_4']['bias'] += 6.0
    return dict(g=g, frames=frames, t_hr=t_hr, geo=geo, A=A, target=target, sigma=sigma, rt=rt, pred=pred, params=params,
                movie=movie)


def _render(s):
    from bhnerf_amd import network, units
    imgs = network.image_plane_prediction(s['params'], s['pred'].apply, s['t_hr'], *s['rt'].values(), units.hr)
    return imgs.detach()


def test_loss_fn_eht_at_config4_size_vs_oracle(dev, setup):
    from bhnerf_amd import engine, observation
    from oracle import oracle_np as onp
    from test_gpu_eht import _ref_loss
    s = setup
    images = _render(s)                                                    # (8, 256, 256) float32 on the device
    assert images.shape == (NT, NPIX, NPIX) and float(images.sum()) > 0
    host = images.cpu().numpy().astype(np.float64)
    flat = images.reshape(NT, -1)
    # 'vis' and 'amp' on the 28 baselines, 'cphase' on 20 triangles of the 8 stations
    tri = observation.closure_triangles(8)[:20]
    pairs = {tuple(p): i for i, p in enumerate(s['g']['pairs'])}
    idx = np.array([[pairs[(a, b)], pairs[(b, c)], pairs[(a, c)]] for a, b, c in tri])          # (20, 3)
    A3 = np.stack([s['A'][:, idx[:, k]] for k in range(3)], axis=1)                                # (8, 3, 20, R)
    A3[:, 2] = np.conj(A3[:, 2])                                                                   # leg c->a
    cp_true = np.angle(np.prod(np.einsum('tckp,tp->tck', A3.astype(np.complex128), s['movie'].reshape(NT, -1).astype(np.complex128)), axis=1))
    cases = {'vis': (s['A'], s['target'], s['sigma']),
             'amp': (s['A'], np.abs(s['target']).astype(np.float32), s['sigma']),
             'cphase': (A3, cp_true.astype(np.float32), np.full(cp_true.shape, 0.1, dtype=np.float32))}
    for dtype, (A, target, sigma) in cases.items():
        loss, dimg = engine.chi2_eht(flat, A, target, sigma, 1.0, dtype)
        ref = onp.loss_eht(host, target.astype(np.complex128 if dtype == 'vis' else np.float64), sigma.astype(np.float64),
                           A.astype(np.complex128), 1.0, dtype)
        assert abs(loss.item() - ref) <= 2e-5 * abs(ref), (dtype, loss.item(), ref)                 # f32 tolerance of north_star
        ref_loss, ref_grad = _ref_loss(host, A, target, sigma, 1.0, dtype)
        assert abs(ref_loss - ref) <= 1e-6 * abs(ref)                     # the two float64 references agree
        err = np.abs(dimg.cpu().numpy().reshape(ref_grad.shape) - ref_grad).max() / np.abs(ref_grad).max()
        assert err < 2e-5, (dtype, err)
        # fixed-order reductions: loss and gradient are bitwise reproducible, and linear in the loss scale
        loss2, dimg2 = engine.chi2_eht(flat, A, target, sigma, 1.0, dtype)
        assert torch.equal(loss, loss2) and torch.equal(dimg, dimg2)
        loss3, dimg3 = engine.chi2_eht(flat, A, target, sigma, 2.0, dtype)
        assert abs(loss3.item() - 2 * loss.item()) <= 1e-6 * abs(loss3.item())
        assert float((dimg3 - 2 * dimg).abs().max()) <= 1e-6 * float(dimg3.abs().max())


def test_eht_visibility_kernels_bandwidth(dev, setup):
    """HBM-bound: A (N, C, nvis, R) complex64 is read once by the GEMV and once by its adjoint; algorithmic bytes
    2 * 8 * N * C * nvis * R (simple_kernels.hip).  With 8 x 28 rows the R axis is split so that the grid fills the chip."""
    from bhnerf_amd import engine
    s = setup
    flat = _render(s).reshape(NT, -1)
    A = torch.as_tensor(s['A'], device=dev)
    tgt = torch.as_tensor(s['target'], device=dev)
    sig = torch.as_tensor(s['sigma'], device=dev)
    for _ in range(3):
        engine.chi2_eht(flat, A, tgt, sig, 1.0, 'vis')
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); engine.chi2_eht(flat, A, tgt, sig, 1.0, 'vis'); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    nbytes = 2 * 8 * NT * 1 * 28 * NPIX * NPIX
    gbs = nbytes / (ms * 1e-3) / 1e9
    print('chi2_eht vis, N=%d nvis=28 R=%d: %.3f ms, %.0f GB/s of %.1f MB algorithmic' % (NT, NPIX * NPIX, ms, gbs, nbytes / 1e6))
    assert gbs > 800.0, gbs          # one block per row (round 1) left 224 blocks on 256 CUs: ~0.3 TB/s


def test_trainstep_eht_arrays_at_config4_size(dev, setup):
    from bhnerf_amd import engine, network, optimization, units
    s = setup
    step = optimization.TrainStep.eht_arrays(s['t_hr'] * units.hr, s['target'], s['sigma'], s['A'], dtype='vis')
    hp = {'num_iters': 10, 'lr_init': 1e-4, 'lr_final': 1e-5, 'seed': 5}
    finals = []
    for rep in range(2):
        opt = optimization.Optimizer(hp, s['pred'], s['rt'])
        with torch.no_grad():
            opt.state.flat.copy_(s['params'].flat)
        idx = np.arange(NT)
        loss0, state, imgs = step(opt.state, s['rt'], idx)
        assert imgs.shape == (1, NT, NPIX, NPIX) and state.step == 1
        if rep == 0:
            direct, _ = engine.chi2_eht(_render(s).reshape(NT, -1), s['A'], s['target'], s['sigma'], 1.0, 'vis', want_grad=False)
            assert abs(float(loss0.sum()) - direct.item()) <= 1e-4 * abs(direct.item())
        finals.append((state.flat.clone(), state.grad[:state.flat.numel()].clone()))
    # the whole step is bitwise reproducible: ray sums are combined per workgroup tile in a fixed order (<= 2 atomics per
    # pixel for rays of <= 129 samples in f32 mode, RaySum in fused_common.h), chi^2 / visibilities / dW in fixed order
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][0], finals[1][0])
    assert float((finals[0][0] - s['params'].flat).abs().max()) > 0


def test_config4_network_4x256_bf16_against_f32_and_oracle(dev, setup):
    """SURVEY 8's table quotes config 4 with the 4x256 network in bf16 arithmetic (the fixture above is 4x128 f32): the
    visibility chi-square and its parameter gradient on the same 256x256x100 geometry, 8 frames -- the bf16 step against
    the f32 step of the same network (images 1e-2 of the maximum, loss 3e-2, gradient 5e-2 relative L2: the bf16 bounds
    of tests/test_gpu_backward.py), the f32 image chi-square against the NumPy oracle's loss_eht, bitwise reproducibility
    of the bf16 step, and TrainStep.eht_arrays moving the parameters."""
    from bhnerf_amd import engine, network, optimization, units
    from oracle import oracle_np as onp
    s = setup
    res = {}
    for mode in ('f32', 'bf16'):
        pred = network.NeRF_Predictor(FOV_M / 2, 2.0, FOV_M / 2, 4.0, net_depth=4, net_width=256, mode=mode, device=dev)
        params = pred.init_params(s['rt'], seed=7)
        with torch.no_grad():
            tree = pred.engine().unflatten(params.flat)
            tree['MLP_0']['Dense_4']['bias'] += 6.0
        flat = params.flat.detach().clone().requires_grad_(True)
        ptree = network.ParamTree(); ptree.flat = flat
        loss, [images] = network.loss_fn_eht(ptree, pred.apply, s['target'], s['sigma'], s['A'], s['t_hr'], *s['rt'].values(), 1.0, units.hr, 'vis')
        loss.backward()
        res[mode] = (float(loss), images.detach().clone(), flat.grad.detach().clone(), pred, params)
        if mode == 'f32':      # the image-domain chi-square of the rendered frames against the oracle
            ref = onp.loss_eht(images.detach().cpu().numpy().astype(np.float64).reshape(NT, NPIX, NPIX), s['target'].astype(np.complex128),
                               s['sigma'].astype(np.float64), s['A'].astype(np.complex128), 1.0, 'vis')
            assert abs(float(loss) - ref) <= 2e-5 * abs(ref)
    (l32, i32, g32, _, _), (l16, i16, g16, pred16, params16) = res['f32'], res['bf16']
    assert float(i32.max()) > 0
    assert float((i16 - i32).abs().max()) <= 1e-2 * float(i32.abs().max())
    assert abs(l16 - l32) <= 3e-2 * abs(l32)
    assert float((g16 - g32).norm() / g32.norm()) <= 5e-2
    # the bf16 training step through the reference-shaped API: reproducible bit for bit, and it moves the parameters
    step = optimization.TrainStep.eht_arrays(s['t_hr'] * units.hr, s['target'], s['sigma'], s['A'], dtype='vis')
    finals = []
    for rep in range(2):
        opt = optimization.Optimizer({'num_iters': 10, 'lr_init': 1e-4, 'lr_final': 1e-5, 'seed': 7}, pred16, s['rt'])
        with torch.no_grad():
            opt.state.flat.copy_(params16.flat)
        loss0, state, imgs = step(opt.state, s['rt'], np.arange(NT))
        assert imgs.shape == (1, NT, NPIX, NPIX) and state.step == 1
        assert abs(float(loss0.sum()) - l16) <= 1e-4 * abs(l16)
        finals.append((state.flat.clone(), state.grad[:state.flat.numel()].clone()))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])
    assert float((finals[0][0] - params16.flat).abs().max()) > 0
