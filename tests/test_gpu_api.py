"""GPU tests of the reference-shaped driver API: Optimizer.run / checkpoints / total_movie_loss /
sample_3d_grid / 'lc' light-curve fitting / generic predictor callables -- the usage contract of the
tutorials and scripts on the HIP engine."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def problem(dev):
    from bhnerf_amd import constants, network, synthetic, units
    H = W = 16; G = 32; nt = 6
    geo = synthetic.synthetic_geodesics(H, W, G, fov_M=16.0, seed=7, S=3)
    t_frames = np.linspace(0, 0.6, nt)
    rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'],
                                      Sigma=geo['Sigma'], t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'],
                                 0.0 * units.hr, J=geo['J'])
    movie = synthetic.hotspot_movie(geo, t_frames, constants.GM_c3('hr'))                  # (nt,H,W) Stokes I
    stokes = np.stack([movie * float(np.mean(geo['J'][s] / geo['J'][0])) for s in range(3)], axis=1)
    return dict(geo=geo, rt=rt, t_frames=t_frames, movie=stokes, H=H, W=W, nt=nt)


def test_lightcurve_fit_decreases_loss_and_checkpoints_resume(dev, problem, tmp_path):
    """scripts/Fit_*.py flow: TrainStep.image(dtype='lc') with Stokes targets, Optimizer.run, checkpoint,
    resume from the checkpoint directory, total_movie_loss with frames."""
    from bhnerf_amd import network, optimization, units
    p = problem
    lc = p['movie'].sum(axis=(-1, -2))                                                     # (nt, 3) light curves
    pred = network.NeRF_Predictor(8.0, 1.0, 8.0, 4.0, net_depth=4, net_width=64, mode='f32', device=dev)
    step = optimization.TrainStep.image(p['t_frames'] * units.hr, lc, sigma=float(np.abs(lc).mean()) * 0.1, dtype='lc')
    ckpt = str(tmp_path / 'run')
    seen = []
    opt = optimization.Optimizer({'num_iters': 30, 'lr_init': 2e-3, 'lr_final': 2e-4, 'seed': 1}, pred, p['rt'],
                                 save_period=10, checkpoint_dir=ckpt, keep=2)
    first = optimization.total_movie_loss(3, opt.state, step, p['rt'])
    opt.run(3, step, p['rt'], log_fns=[optimization.LogFn(lambda o: seen.append((o.step, float(torch.as_tensor(o.loss).mean()))), 10)])
    last, frames = optimization.total_movie_loss(3, opt.state, step, p['rt'], return_frames=True)
    assert opt.state.step == 30 and [s for s, _ in seen] == [1, 10, 20, 30]
    assert last < 0.7 * first, (first, last)
    assert frames.shape == (p['nt'], 3, p['H'], p['W']) and np.isfinite(frames).all()
    import os
    assert sorted(os.listdir(ckpt)) == ['NeRF_Predictor_params.yml', 'checkpoint_20', 'checkpoint_30']    # keep=2
    # resume: a new Optimizer on the same directory restores step, parameters and Adam moments
    pred2 = network.NeRF_Predictor.from_yml(ckpt, mode='f32', device=dev)
    opt2 = optimization.Optimizer({'num_iters': 5, 'lr_init': 2e-3, 'lr_final': 2e-4}, pred2, p['rt'], checkpoint_dir=ckpt)
    assert opt2.state.step == 30 and torch.equal(opt2.state.flat, opt.state.flat) and torch.equal(opt2.state.m, opt.state.m)
    opt2.run(3, step, p['rt'])
    assert opt2.state.step == 35 and opt2.init_step == 31
    # the files are flax TrainState msgpack state dicts (optimization.py:118-121): step, params tree, optax.adam state
    from bhnerf_amd import checkpoints
    sd = checkpoints.restore_checkpoint(os.path.join(ckpt, 'checkpoint_30'), None)
    assert int(sd['step']) == 30 and int(sd['opt_state']['0']['count']) == 30 and int(sd['opt_state']['1']['count']) == 30
    assert sorted(sd['params']['MLP_0']) == ['Dense_%d' % i for i in range(5)]
    k0 = sd['params']['MLP_0']['Dense_0']['kernel']
    assert k0.shape == (21, 64) and k0.dtype == np.float32
    assert np.array_equal(k0, opt.state.params['MLP_0']['Dense_0']['kernel'].cpu().numpy())
    assert sd['opt_state']['0']['nu']['MLP_0']['Dense_3']['kernel'].shape == (64 + 21, 64)
    # image-plane movie and 3-D samples straight from the checkpoint directory (network.py:842-848, 896-906)
    movie = network.image_plane_checkpoint(p['rt'], ckpt, p['t_frames'] * units.hr, batchsize=4, mode='f32', device=dev)
    _, frames35 = optimization.total_movie_loss(3, opt2.state, step, p['rt'], return_frames=True)
    assert movie.shape == frames35.shape and np.allclose(movie, frames35, rtol=1e-6, atol=0)
    vol = network.sample_checkpoint_3d(ckpt, fov=10.0, resolution=8, mode='f32', device=dev)
    assert np.array_equal(vol, network.sample_3d_grid(pred2.apply, opt2.state.params, fov=10.0, resolution=8))
    # a checkpoint written by a round-1 build (torch.save of the flat buffers) is still restored
    legacy = tmp_path / 'legacy'
    legacy.mkdir()
    torch.save({'step': 7, 'params': opt.state.flat.cpu(), 'm': opt.state.m.cpu(), 'v': opt.state.v.cpu()}, str(legacy / 'checkpoint_7'))
    st = pred2.init_state(pred2.init_params(p['rt']), checkpoint_dir=str(legacy))
    assert st.step == 7 and torch.equal(st.flat, opt.state.flat) and torch.equal(st.v, opt.state.v)
    with pytest.raises(AttributeError):
        optimization.TrainStep.image(p['t_frames'] * units.hr, lc, dtype='nope')(opt.state, p['rt'], np.arange(3))


def test_sample_3d_grid_and_generic_predictor_callable(dev, problem):
    from bhnerf_amd import network, units
    from oracle import oracle_np as onp
    p = problem
    pred = network.NeRF_Predictor(8.0, 0.0, 8.0, 100.0, net_depth=4, net_width=64, mode='f32', device=dev)
    params = pred.init_params(p['rt'], seed=2)
    vol = network.sample_3d_grid(pred.apply, params, fov=10.0, resolution=8)
    assert vol.shape == (8, 8, 8) and (vol > 0).any()
    vol2 = network.sample_3d_grid(pred.apply, params, fov=10.0, resolution=8, chunk=4)
    assert np.array_equal(vol, vol2)
    # oracle on the same grid (t=0, Omega=0 -> no warp): float64 restatement with the same weights
    grid = np.linspace(-5, 5, 8)
    coords = np.array(np.meshgrid(grid, grid, grid, indexing='ij'))
    tree = {'MLP_0': {k: {kk: vv.cpu().numpy().astype(np.float64) for kk, vv in v.items()} for k, v in params['MLP_0'].items()}}
    ref = onp.predictor_apply(tree, 0.0, coords, 0.0, 0.0, 0.0, 0.0, GM_c3=1.0, scale=8.0, rmin=0.0, rmax=8.0, z_width=100.0)
    assert np.abs(vol - ref).max() < 1e-5 * ref.max()
    with pytest.raises(AttributeError):
        network.sample_3d_grid(pred.apply, params)
    # any callable can stand in for the predictor: emission is integrated by the stand-alone kernel
    e_fix = torch.rand((p['nt'],) + p['geo']['Omega'].shape, device=dev)
    images = network.image_plane_prediction(None, lambda v, *a: e_fix, p['t_frames'], *p['rt'].values(), units.hr)
    geo = p['geo']
    ref = onp.image_plane_prediction(e_fix.cpu().numpy().astype(np.float64), geo['J'].astype(np.float64), geo['g'].astype(np.float64),
                                     geo['dtau'].astype(np.float64), geo['Sigma'].astype(np.float64))
    assert tuple(images.shape) == ref.shape and np.abs(images.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()


def test_two_losses_take_sequential_adam_steps(dev, problem):
    """TrainStep.__add__: each loss takes its own Adam step (optimization.py:175-178), they are not summed."""
    from bhnerf_amd import network, optimization, units
    p = problem
    pred = network.NeRF_Predictor(8.0, 0.0, 8.0, 4.0, net_depth=4, net_width=32, mode='f32', device=dev)
    full = optimization.TrainStep.image(p['t_frames'] * units.hr, p['movie'], dtype='full')
    lc = optimization.TrainStep.image(p['t_frames'] * units.hr, p['movie'].sum(axis=(-1, -2)), dtype='lc', scale=0.5)
    both = full + lc
    state = pred.init_state(pred.init_params(p['rt']), num_iters=10)
    loss, state, images = both(state, p['rt'], np.array([0, 2, 4]))
    assert state.step == 2 and loss.shape == (1,) and images.shape == (1, 3, 3, p['H'], p['W'])


def test_training_on_traced_kerr_geodesics(dev):
    """f3 end to end: own ray tracer -> Doppler factor and Stokes factors -> raytracing_args -> polarised light-curve
    fit steps on the HIP engine (the flow of scripts/Fit_*: kgeo.image_plane_geos, doppler_factor, parallel_transport)."""
    from bhnerf_amd import kgeo, network, optimization, units
    geos = kgeo.image_plane_geos(0.5, np.deg2rad(30.0), (-8.0, 8.0), (-8.0, 8.0), ngeo=32, num_alpha=8, num_beta=8)
    Omega = 1.0 / (geos.r ** 1.5 + geos.spin)
    umu = kgeo.azimuthal_velocity_vector(geos, Omega)
    geos['g'] = kgeo.doppler_factor(geos, umu)
    b = kgeo.magnetic_field_fluid_frame(geos, umu, arad=0.0, avert=1.0, ator=0.0)
    J = np.nan_to_num(kgeo.parallel_transport(geos, umu, geos['g'], b, Q_frac=0.5, V_frac=0), posinf=0.0, neginf=0.0)
    assert J.shape == (3, 8, 8, 32) and np.isfinite(geos['g']).all()
    rt = network.raytracing_args(geos, np.nan_to_num(Omega), 0.0, 0.0 * units.hr, J=J)
    t_frames = np.linspace(0.0, 0.5, 4) * units.hr
    pred = network.NeRF_Predictor(8.0, 2.0, 8.0, 4.0, net_depth=4, net_width=64, mode='f32', device=dev)
    target = np.abs(np.random.default_rng(0).standard_normal((4, 3))) * 1e-3
    step = optimization.TrainStep.image(t_frames, target, sigma=1e-3, dtype='lc')
    opt = optimization.Optimizer({'num_iters': 6, 'lr_init': 1e-3, 'lr_final': 1e-4}, pred, rt)
    first = optimization.total_movie_loss(2, opt.state, step, rt)
    opt.run(2, step, rt)
    last, frames = optimization.total_movie_loss(2, opt.state, step, rt, return_frames=True)
    assert np.isfinite(first) and np.isfinite(last) and frames.shape == (4, 3, 8, 8) and np.isfinite(frames).all()
    assert opt.state.step == 6


def test_alma_chi2_from_checkpoint(dev, tmp_path):
    """alma.get_raytracing_args -> Optimizer (polarised 'lc' fit) -> alma.chi2_lightcurves / chi2_df (alma.py:66-117)."""
    import os
    from bhnerf_amd import alma, network, optimization, units
    params = dict(num_alpha=8, num_beta=8, fov_M=16.0, z_width=4.0, rmin='ISCO', Q_frac=0.5,
                  b_consts=dict(arad=0.0, avert=1.0, ator=0.0), Omega_dir='cw', t_start_obs=9.5)
    rt = alma.get_raytracing_args(np.deg2rad(25.0), 0.2, params)
    t = (9.5 + np.linspace(0.0, 0.5, 5)) * units.hr
    data = np.abs(np.random.default_rng(3).standard_normal((5, 3))) * 1e-2
    ckpt = str(tmp_path / 'inc_25.0_seed_0')
    pred = network.NeRF_Predictor(8.0, 2.0, 8.0, 4.0, net_depth=4, net_width=64, mode='f32', device=dev)
    step = optimization.TrainStep.image(t, data, sigma=1e-2, dtype='lc')
    opt = optimization.Optimizer({'num_iters': 4, 'lr_init': 1e-3, 'lr_final': 1e-4}, pred, rt, save_period=4, checkpoint_dir=ckpt)
    opt.run(5, step, rt)
    assert os.path.exists(os.path.join(ckpt, 'checkpoint_4'))
    chi2 = alma.chi2_lightcurves(rt, ckpt, t, data, sigma=1e-2, batchsize=5, mode='f32', device=dev)
    _, frames = optimization.total_movie_loss(5, opt.state, step, rt, return_frames=True)
    want = np.sum(((frames.sum(axis=(-1, -2)) - data) / 1e-2) ** 2) / 5
    assert np.isfinite(chi2) and abs(chi2 - want) <= 1e-5 * abs(want)
    df = alma.chi2_df([25.0, 35.0], 0.2, [0], params, str(tmp_path / 'inc_{}_seed_{}'), t, data, sigma=1e-2, final_step=4)
    assert np.isnan(df.loc[35.0, 'seed 0']) and np.isfinite(df.loc[25.0, 'seed 0'])


def test_summary_writer_log_functions(dev, problem, tmp_path):
    """SummaryWriter.recovery_3d / plot_lc_datafit as used by the fit scripts (optimization.py:310-347)."""
    import json
    import types
    from bhnerf_amd import network, optimization, units
    p = problem
    lc = p['movie'].sum(axis=(-1, -2))
    pred = network.NeRF_Predictor(8.0, 1.0, 8.0, 4.0, net_depth=4, net_width=64, mode='f32', device=dev)
    step = optimization.TrainStep.image(p['t_frames'] * units.hr, lc, sigma=float(np.abs(lc).mean()) * 0.1, dtype='lc')
    opt = optimization.Optimizer({'num_iters': 4, 'lr_init': 1e-3, 'lr_final': 1e-4}, pred, p['rt'])
    writer = optimization.SummaryWriter(str(tmp_path / 'tb'))
    ax = np.linspace(-5.0, 5.0, 6)
    truth = types.SimpleNamespace(x=ax, y=ax, z=ax, data=np.full((6, 6, 6), 0.1), shape=(6, 6, 6))   # xarray-like volume
    opt.run(3, step, p['rt'], log_fns=[optimization.LogFn(writer.recovery_3d(10.0, emission_true=truth), 2),
                                       optimization.LogFn(writer.recovery_3d(10.0, vis_res=8), 4)])
    writer.plot_lc_datafit(opt, 'stokes', step, lc, ['I', 'Q', 'U'], p['t_frames'], batchsize=3)
    writer.close()
    if writer._tb is None:
        recs = [json.loads(l) for l in open(tmp_path / 'tb' / 'scalars.jsonl')]
        tags = [(r['tag'], r['step']) for r in recs]
        assert ('emission/mse', 1) in tags and ('emission/psnr', 4) in tags and ('datafit/stokes', 4) in tags
        assert np.load(tmp_path / 'tb' / 'emission_estimate_4.npy').shape in ((8, 3, 8, 8), (6, 3, 6, 6))
        assert (tmp_path / 'tb' / 'lightcurve_stokes_4.png').stat().st_size > 1000


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_edge_cases_empty_domain_pre_injection_and_single_sample_ray(dev, mode):
    """Nothing to render (every sample outside the recovery domain, emission.py:370-373; every sample before the injection
    time, emission.py:204-205 -> NaN -> 0, network.py:226-232) and the smallest possible geometry (one ray, two samples)."""
    from bhnerf_amd import network, optimization, synthetic, units
    geo = synthetic.synthetic_geodesics(8, 8, 40, S=3, seed=1)
    as_geos = lambda g: dict(x=g['coords'][0], y=g['coords'][1], z=g['coords'][2], dtau=g['dtau'], Sigma=g['Sigma'], t=g['t_geos'], g=g['g'])
    rt = network.raytracing_args(as_geos(geo), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=geo['J'])
    t = np.linspace(0, 1, 4) * units.hr
    step = optimization.TrainStep.image(t, np.full((4, 3), 1e-3), sigma=1e-3, dtype='lc')
    # (i) empty domain: zero images, chi^2 of the bare targets, zero gradient -> Adam leaves the parameters alone
    pred = network.NeRF_Predictor(8.0, 100.0, 200.0, 4.0, net_depth=4, net_width=64, mode=mode, device=dev)
    opt = optimization.Optimizer({'num_iters': 3, 'lr_init': 1e-3, 'lr_final': 1e-4}, pred, rt)
    before = opt.state.flat.clone()
    opt.run(2, step, rt)
    loss, frames = optimization.total_movie_loss(2, opt.state, step, rt, return_frames=True)
    assert loss == pytest.approx(3.0) and np.abs(frames).max() == 0.0 and torch.equal(opt.state.flat, before)
    # (ii) all samples precede the injection: same, and nothing turns NaN
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=64, mode=mode, device=dev)
    late = dict(rt)
    late['t_injection'] = 1e6
    opt = optimization.Optimizer({'num_iters': 3, 'lr_init': 1e-3, 'lr_final': 1e-4}, pred, late)
    opt.run(2, step, late)
    loss, frames = optimization.total_movie_loss(2, opt.state, step, late, return_frames=True)
    assert loss == pytest.approx(3.0) and np.abs(frames).max() == 0.0 and bool(torch.isfinite(opt.state.flat).all())
    # (iii) one frame, one ray, two samples
    g1 = synthetic.synthetic_geodesics(1, 1, 2, seed=1)
    rt1 = network.raytracing_args(as_geos(g1), g1['Omega'], g1['t_injection'], 0.0 * units.hr, J=1.0)
    step1 = optimization.TrainStep.image(np.array([0.3]) * units.hr, np.full((1, 1, 1), 1e-3), sigma=1e-3, dtype='full')
    opt = optimization.Optimizer({'num_iters': 2, 'lr_init': 1e-3, 'lr_final': 1e-4}, pred, rt1)
    opt.run(1, step1, rt1)
    assert np.isfinite(float(np.mean(opt.loss))) and opt.state.step == 2


def test_geometry_cache_sees_in_place_edits(dev, problem):
    """NeRF_Predictor.geometry caches the prepared ray geometry per set of ray-tracing arrays; the key holds a fingerprint of
    the CONTENTS (NumPy: strided sample; tensors: version counter), so an in-place edit of an array gives a new geometry
    instead of silently re-using the stale one (round 2: keyed on id() only)."""
    from bhnerf_amd import network
    geo = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in problem['geo'].items()}
    pred = network.NeRF_Predictor(8.0, 1.0, 8.0, 4.0, net_depth=4, net_width=64, mode='f32', device=dev)
    args = (geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    g1 = pred.geometry(*args)
    assert pred.geometry(*args) is g1                                  # unchanged arrays: cache hit
    w_before = g1.w.clone()
    geo['g'] *= 2.0                                                    # in place: same object, new contents
    g2 = pred.geometry(*args)
    assert g2 is not g1
    assert torch.allclose(g2.w, 4.0 * w_before)                        # w = g^2 dtau Sigma
    t = torch.as_tensor(np.ascontiguousarray(geo['Omega'], dtype=np.float32), device=dev)
    args_t = (geo['coords'], t, geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    g3 = pred.geometry(*args_t)
    assert pred.geometry(*args_t) is g3
    t.mul_(0.5)                                                        # tensors: exact, through the version counter
    assert pred.geometry(*args_t) is not g3


@pytest.mark.parametrize('case', ['lc_stokes_128', 'full_256_masked', 'general_posenc6'])
def test_hip_graph_step_is_bitwise_equal_to_the_eager_step(dev, problem, case):
    """hparams['hip_graph']: the training step captured into a HIP graph (optimization.GraphedImageStep: frame indices, tM0 and
    Adam's scalars in device buffers) leaves bitwise the same parameters, Adam moments and losses as the eager step, over
    several steps with different frame batches -- for the fused width-128 backward and for the generic tape path.  The general
    path (posenc_deg 6: csrc/general_mlp.hip) is captured the same way; its pixels take one float atomic per 32-point group of a
    ray, so two runs agree to rounding, not bitwise."""
    from bhnerf_amd import network, optimization, units
    p = problem
    if case == 'lc_stokes_128':
        tgt, dtype, width, dom = p['movie'].sum(axis=(-1, -2)), 'lc', 128, (8.0, 0.0, np.inf, np.inf)
    elif case == 'general_posenc6':
        tgt, dtype, width, dom = p['movie'].sum(axis=(-1, -2)), 'lc', 96, (8.0, 0.0, np.inf, np.inf)
    else:
        tgt, dtype, width, dom = p['movie'][:, 0], 'full', 256, (8.0, 1.0, 8.0, 4.0)
    rt = p['rt'] if case != 'full_256_masked' else dict(p['rt'], J=1.0)
    deg = 6 if case == 'general_posenc6' else 3
    runs = {}
    for graph in (False, True):
        pred = network.NeRF_Predictor(*dom, posenc_deg=deg, net_depth=4, net_width=width, mode='bf16', device=dev)
        step = optimization.TrainStep.image(p['t_frames'] * units.hr, tgt, sigma=float(np.abs(tgt).mean()) * 0.1, dtype=dtype)
        opt = optimization.Optimizer({'num_iters': 7, 'lr_init': 2e-3, 'lr_final': 2e-4, 'seed': 1, 'hip_graph': graph}, pred, rt)
        losses = []
        init = opt.state.flat.clone()
        opt.run(2, step, rt, log_fns=[optimization.LogFn(lambda o: losses.append(torch.as_tensor(o.loss).clone()))])
        assert opt.state.step == 7
        assert bool(step._graphs) == graph and (not graph or all(step._graphs.values()))
        runs[graph] = (opt.state.flat.clone(), opt.state.m.clone(), opt.state.v.clone(), torch.stack([l.reshape(-1)[0] for l in losses]))
    if case == 'general_posenc6':
        assert torch.allclose(runs[False][3], runs[True][3], rtol=1e-4) and float((runs[False][0] - runs[True][0]).abs().max()) < 1e-3
    else:
        for a, b in zip(runs[False], runs[True]):
            assert torch.equal(a, b)
    assert not torch.equal(runs[True][0], init)                      # (the steps did move the parameters)


def test_hip_graph_step_with_the_8bit_tape(dev, problem):
    """mode='bf16_t8' under hparams['hip_graph']: the tape-scale kernels (t8_prepare / t8_update, fused_bwd.hip) are part of the
    captured backward and keep their state in the workspace, so a replayed step follows the eager one.  Not bitwise: the
    warm-up steps of the capture leave other scale ratios behind than the eager run's history (powers of two that almost
    always coincide).  Compared on the LOSS curve of seven Adam steps: Adam turns every gradient entry into a step of size lr,
    so parameters with near-zero gradients move with the sign of the rounding noise in any reduced-precision mode."""
    from bhnerf_amd import network, optimization, units
    p = problem
    tgt, dom, rt = p['movie'][:, 0], (8.0, 1.0, 8.0, 4.0), dict(p['rt'], J=1.0)
    runs = {}
    for mode, graph in (('bf16', False), ('bf16_t8', False), ('bf16_t8', True)):
        pred = network.NeRF_Predictor(*dom, net_depth=4, net_width=256, mode=mode, device=dev)
        step = optimization.TrainStep.image(p['t_frames'] * units.hr, tgt, sigma=float(np.abs(tgt).mean()) * 0.1, dtype='full')
        opt = optimization.Optimizer({'num_iters': 7, 'lr_init': 2e-3, 'lr_final': 2e-4, 'seed': 1, 'hip_graph': graph}, pred, rt)
        losses = []
        opt.run(2, step, rt, log_fns=[optimization.LogFn(lambda o: losses.append(float(torch.as_tensor(o.loss).reshape(-1)[0])))])
        assert opt.state.step == 7 and bool(step._graphs) == graph
        assert torch.isfinite(opt.state.flat).all()
        runs[(mode, graph)] = np.array(losses)
    ref = runs[('bf16', False)]
    assert len(ref) >= 3 and ref[-1] < ref[0]
    for k in (('bf16_t8', False), ('bf16_t8', True)):
        assert np.abs(runs[k] - ref).max() < 0.03 * ref.max(), (k, runs[k], ref)
    assert np.abs(runs[('bf16_t8', True)] - runs[('bf16_t8', False)]).max() < 0.01 * ref.max()


def test_hip_graph_is_never_replayed_on_stale_addresses_or_geometry(dev, problem):
    """A HIP graph bakes device addresses in: the captured step must be dropped and captured again when (a) a
    second ray set with MORE points makes the engine re-allocate its workspace, (b) a ray-tracing dict is edited in place
    (new geometry), (c) a new dict is allocated where a dead one lived (recycled id()), (d) the geometry cache is cleared.
    Every scenario is run eagerly and through the graph path: parameters, Adam moments and losses bitwise equal."""
    import gc
    from bhnerf_amd import network, optimization, synthetic, units
    p = problem
    tgt = p['movie'].sum(axis=(-1, -2))
    dom = (8.0, 0.0, np.inf, np.inf)

    def make_rt(H, seed, J=True):
        geo = synthetic.synthetic_geodesics(H, H, 32, fov_M=16.0, seed=seed, S=3)
        return network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                            t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=geo['J'])

    def run(graph):
        pred = network.NeRF_Predictor(*dom, net_depth=4, net_width=128, mode='bf16', device=dev)
        step = optimization.TrainStep.image(p['t_frames'] * units.hr, tgt, sigma=float(np.abs(tgt).mean()) * 0.1, dtype='lc')
        step.use_graph = graph
        rt_small, rt_big = make_rt(12, 3), make_rt(20, 4)
        opt = optimization.Optimizer({'num_iters': 50, 'lr_init': 2e-3, 'lr_final': 2e-4, 'seed': 1}, pred, rt_small)
        st, losses, events = opt.state, [], []

        def do(rt, idx):
            nonlocal st
            loss, st, _ = step(st, rt, np.asarray(idx))
            losses.append(loss.reshape(-1)[0].clone())

        do(rt_small, [0, 1]); do(rt_small, [2, 3])
        ws0 = pred.engine()._ws
        do(rt_big, [1, 4])                                   # (a) larger tape: the workspace is re-allocated
        events.append(pred.engine()._ws is not ws0)
        do(rt_small, [0, 5]); do(rt_small, [3, 2])           # ... the small set's graph was captured on the old workspace
        rt_small['g'] *= 1.5                                 # (b) in-place edit: new geometry
        do(rt_small, [0, 1])
        for k in range(4):                                   # (c) dicts that die and are re-born, possibly at the same id()
            rt_tmp = make_rt(12, 10 + k)
            do(rt_tmp, [k, k + 1])
            del rt_tmp
            gc.collect()
        pred.clear_geometry_cache()                          # (d)
        do(rt_big, [2, 3])
        return st.flat.clone(), st.m.clone(), st.v.clone(), torch.stack(losses), events, step

    eager, graphed = run(False), run(True)
    assert eager[4] == [True] and graphed[4] == [True]       # the scenario really re-allocated the workspace
    assert any(graphed[5]._graphs.values())
    for a, b in zip(eager[:4], graphed[:4]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('width', [128, 256])
def test_recorded_tape_can_be_replayed(dev, problem, width):
    """bhn_render_bwd_tape only reads what the training forward recorded (the fused width-128 backward used to turn
    the tape's emission into dout in place): a second call on the same tape returns the same gradient bit for bit, and a
    call with other dimages in between does not disturb it."""
    from bhnerf_amd import constants, engine, network
    geo = problem['geo']
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=width, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], geo['J'], geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(4, width).init(1, 21)))
    tM0 = engine.frame_offsets(problem['t_frames'][:4], 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    d1 = torch.randn((4, geom.Sx, geom.R), device=dev, generator=gen)
    d2 = torch.randn((4, geom.Sx, geom.R), device=dev, generator=gen)
    eng.render_train(geom, tM0)
    g1 = eng.render_bwd_tape(geom, tM0, d1).clone()
    g2 = eng.render_bwd_tape(geom, tM0, d2).clone()
    g1b = eng.render_bwd_tape(geom, tM0, d1).clone()
    assert float(g1.abs().max()) > 0 and not torch.equal(g1, g2)
    assert torch.equal(g1, g1b)


@pytest.mark.parametrize('width', [128, 256])
def test_clock_probe_stamps_and_tape_info(dev, problem, width):
    """ABI 5's measurement aids through the C ABI: `bhn_frames.clock_probe` -- workgroup 0 of each fused MLP kernel stamps
    {s_memtime, s_memrealtime} at its start and end into its own slot, nothing else changes (images and gradients bit-identical
    with and without the probe) -- and `bhn_tape_info` against the workspace the library asks for."""
    from bhnerf_amd import _hip, constants, engine as E, network
    geo = problem['geo']
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=width, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], geo['J'], geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(4, width).init(1, 21)))
    tM0 = E.frame_offsets(problem['t_frames'][:4], 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = torch.rand((4, geom.Sx, geom.R), device=dev) - 0.4
    img0 = eng.render(geom, tM0).clone()
    imt0 = eng.render_train(geom, tM0).clone()
    g0 = eng.render_bwd_tape(geom, tM0, dimg).clone()
    clk = torch.zeros((4 * _hip.BHN_CLK_SLOTS,), dtype=torch.int64, device=dev)
    eng.clock_probe = clk
    try:
        img1 = eng.render(geom, tM0).clone()
        imt1 = eng.render_train(geom, tM0).clone()
        g1 = eng.render_bwd_tape(geom, tM0, dimg).clone()
    finally:
        eng.clock_probe = None
    assert torch.equal(img0, img1) and torch.equal(imt0, imt1) and torch.equal(g0, g1)
    c = clk.cpu().numpy().reshape(_hip.BHN_CLK_SLOTS, 4)
    info = eng.tape_info((geom.P_eff + 31) // 32)
    used = [_hip.BHN_CLK_FWD, _hip.BHN_CLK_FWD_TRAIN, _hip.BHN_CLK_CHAIN] + ([] if info['flags']['fused128'] else [_hip.BHN_CLK_DW])
    for slot in used:
        t0, r0, t1, r1 = (int(v) for v in c[slot])
        assert t1 > t0 and r1 > r0, (slot, c[slot])
        mhz = 100.0 * (t1 - t0) / (r1 - r0)                        # s_memrealtime counts at 100 MHz
        assert 300.0 < mhz < 2600.0, (slot, mhz)
    if info['flags']['fused128']:
        assert not c[_hip.BHN_CLK_DW].any()                        # (no separate dW kernel on the fused 4x128 path)
    # the tape the library lays out for these frames is what bhn_tape_info says it streams (forward + chain writes, rounded up)
    assert info['flags']['fused128'] == (width == 128) and info['flags']['ga0_chain'] == (width == 256)
    groups = 4 * ((geom.P_eff + 31) // 32)
    assert eng.workspace(4, geom.P_eff).numel() >= groups * (info['fwd_write'] + info['chain_write'])


def test_mfma_probe_reports_a_plausible_ceiling(dev):
    """bhn_mfma_probe (bench.py: mfma_peak_this_box): the register-operand bf16 MFMA loop runs, its clock stamps are sane and
    the rate it implies lies between a tenth of and the whole dense bf16 peak."""
    import ctypes as C
    from bhnerf_amd import _hip
    lib = _hip.lib()
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    clk = torch.zeros((2 * ncu,), dtype=torch.int64, device=dev)
    sink = torch.zeros((1024,), dtype=torch.float32, device=dev)
    iters = 4000
    _hip.check(lib.bhn_mfma_probe(ncu, iters, _hip.ptr(clk), _hip.ptr(sink), _hip.stream_ptr(dev)))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _hip.check(lib.bhn_mfma_probe(ncu, iters, _hip.ptr(clk), _hip.ptr(sink), _hip.stream_ptr(dev)))
    b.record(); torch.cuda.synchronize()
    tflops = ncu * 8 * iters * 16 * 32768.0 / (a.elapsed_time(b) * 1e-3) / 1e12
    c = clk.cpu().numpy().reshape(ncu, 2)
    mhz = np.median(100.0 * c[:, 0] / c[:, 1])
    assert (c > 0).all() and 300.0 < mhz < 2600.0 and 250.0 < tflops < 2500.0, (mhz, tflops)
    assert lib.bhn_mfma_probe(0, iters, None, _hip.ptr(sink), _hip.stream_ptr(dev)) == 1       # BHN_EINVAL
