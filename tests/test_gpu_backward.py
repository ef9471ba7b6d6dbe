"""GPU parity tests of the backward half of the hot path (through the C ABI): parameter gradients of
loss_fn_image against (i) torch.autograd on the float64 oracle and (ii) the finite differences of
the REFERENCE's own loss_fn_image stored in the golden fixtures; Adam training steps against the
oracle trainer; workspace frame-chunking.

Tolerances (error / largest gradient entry, and relative L2), set from what the kernels deliver on these fixtures
(tools/diag_tol.py, round 2): f32 mode 2e-5 for both (observed <= 1.6e-6), loss 1e-5 (observed <= 1.6e-6); bf16 mode L2
6e-2, max 1.2e-1, loss 3e-2 (observed <= 3.7e-2 / 7.5e-2 / 1.7e-2: bf16 activations and deltas on the tape, f32
accumulation, and only 100-500 points per fixture to average the rounding over -- the full-size test holds 2e-2).
A ReLU-net gradient is discontinuous: where a pre-activation lies within f32 rounding of zero, f32 and f64 may
disagree on relu' for that (point, unit) and every layer below changes by that one point's contribution.  Such cases
are DETECTED (conftest.relu_tie_count on the float64 reference forward) and ADJUDICATED, not waved through (round 3;
rounds 1-2 fell back to a 1e-3 bound): the gradient is a sum over ray samples, so the same problem is re-run with exactly
the samples that have a tie taken out (their Doppler weight g set to 0 on both sides: they no longer reach the image) and
the f32 gradient must then meet the SAME 2e-5 bounds -- otherwise the disagreement was not a tie and the test fails.  Where
the ties are few, nudging every weight by a random relative 1e-6 ... 1e-3 until none is left does the same (fixtures:
their finite-difference reference is tied to the weights, so only the oracle comparison is repeated).  A random problem
with 14 M pre-activations has ~10^2 of them inside the detection band at ANY weights, which is why the ray samples, not
the weights, are what the random-problem test moves."""
import numpy as np
import pytest
import torch

from conftest import golden_tree, relu_tie_count
from oracle import oracle_np as onp
from oracle import oracle_torch as ot

pytestmark = pytest.mark.gpu
PRED = ['a', 'b', 'c', 'd', 'e', 'f']
GTOL = {'f32': 2e-5, 'bf16': 1.2e-1}       # max-norm
L2TOL = {'f32': 2e-5, 'bf16': 6e-2}        # relative L2
LOSSTOL = {'f32': 1e-5, 'bf16': 3e-2}
RANDOM_PROBLEM_SHAPE = (9, 7, 50, 3)       # rays H x W, samples per ray, frames of test_random_problem (tools/fuzz_parity.py varies it)
RANDOM_PROBLEM_DOMAIN = (8.0, 2.5, 8.0, 4.0)   # scale, rmin, rmax, z_width of the recovery domain (fuzz: random)
RANDOM_PROBLEM_JITTER = (0.0, 0.0, 0.0)    # offsets of the alpha / beta / sample grids: regular grids of other shapes put samples EXACTLY
                                           # on the domain boundary (e.g. 10 x 7 rays x 64 samples: alpha^2 + beta^2 + s^2 = rmax^2),
                                           # where the f32 kernel and the f64 oracle legitimately disagree about the mask
TIE_NUDGES = (1e-6, 1e-5, 1e-4, 1e-3)      # relative weight nudges tried to move detected ReLU ties off zero (module docstring)


def l2err(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def oracle_trainer(g, dtype=torch.float64, **kw):
    hp = g['hparams']
    ks, bs = ot.tree_to_lists(golden_tree(g), dtype)
    t = lambda x: torch.tensor(x, dtype=dtype)
    geom = dict(coords=t(g['coords']), Omega=t(g['Omega']), t_geos=t(g['t_geos']), g=t(g['g']), dtau=t(g['dtau']),
                Sigma=t(g['Sigma']), J=(t(g['J']) if g['J'].ndim else None), t_start_obs=float(g['t_start_obs']),
                t_injection=float(g['t_injection']))
    hpd = dict(GM_c3=onp.GM_C3_SGRA_HR, scale=hp[0], rmin=hp[1], rmax=hp[2], z_width=hp[3], posenc_deg=int(hp[4]),
               net_depth=int(hp[5]))
    return ot.CpuTrainer(ks, bs, geom, hpd, **kw), t


def flat_from_lists(ks, bs):
    """flax tree order: kernel_0, bias_0, kernel_1, ..."""
    return np.concatenate([np.concatenate([k.detach().numpy().ravel(), b.detach().numpy().ravel()]) for k, b in zip(ks, bs)])


def targets(g, dt):
    S = g['J'].shape[0] if g['J'].ndim else None
    b = len(g['t_frames'])
    sp = g['coords'].shape[1:3]
    if dt == 'full':
        shape = (b, S) + sp if S else (b,) + sp
    else:
        shape = (b, S) if S else (b,)
    return {k: g[k + '_' + dt].reshape(shape) for k in ('target', 'sigma', 'offset')}


def device_setup(g, mode, dev):
    from bhnerf_amd import network
    hp = g['hparams']
    pred = network.NeRF_Predictor(hp[0], hp[1], hp[2], hp[3], posenc_deg=int(hp[4]), net_depth=int(hp[5]),
                                  net_width=int(hp[6]), mode=mode, device=dev)
    f = lambda k: np.ascontiguousarray(g[k].astype(np.float32))
    rt = dict(coords=f('coords'), Omega=f('Omega'), J=(f('J') if g['J'].ndim else 1.0), g=f('g'), dtau=f('dtau'),
              Sigma=f('Sigma'), t_start_obs=float(g['t_start_obs']), t_geos=f('t_geos'), t_injection=float(g['t_injection']))
    return pred, rt


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('dt', ['full', 'lc'])
@pytest.mark.parametrize('tag', PRED)
def test_gradient_vs_oracle_and_reference_fd(dev, golden, tag, dt, mode):
    from bhnerf_amd import network, units
    g = golden('g5_predict_' + tag)
    tr, t = oracle_trainer(g)
    tg = targets(g, dt)
    scale = float(g['hparams'][7])
    loss_ref, _, grads_ref = tr.loss_and_grad(t(g['t_frames']), t(tg['target']), t(tg['sigma']), t(tg['offset']), scale, dt)
    n = len(tr.k)
    gref = np.concatenate([np.concatenate([grads_ref[i].numpy().ravel(), grads_ref[n + i].numpy().ravel()]) for i in range(n)])

    pred, rt = device_setup(g, mode, dev)
    params = pred.engine().flatten(golden_tree(g)).requires_grad_(True)
    tree = network.ParamTree()
    tree.flat = params
    loss, [images] = network.loss_fn_image(tree, pred.apply, tg['target'], tg['sigma'], tg['offset'], g['t_frames'],
                                           rt['coords'], rt['Omega'], rt['J'], rt['g'], rt['dtau'], rt['Sigma'],
                                           rt['t_start_obs'], rt['t_geos'], rt['t_injection'], scale, units.hr, dt)
    loss.backward()
    gdev = params.grad.cpu().numpy().astype(np.float64)
    assert abs(loss.item() - loss_ref.item()) <= LOSSTOL[mode] * abs(loss_ref.item())
    gmax = np.abs(gref).max()
    assert gmax > 0
    ties = relu_tie_count(g) if mode == 'f32' else 0          # (fixture b has one pre-activation inside the f32 rounding band)
    gtol, l2tol = GTOL[mode], L2TOL[mode]
    err = np.abs(gdev - gref).max() / gmax
    if mode == 'f32' and ties and not (err < gtol and l2err(gdev, gref) < l2tol):
        # a detected tie only excuses the original if the same fixture with its weights nudged off the tie meets the bounds
        for nudge in TIE_NUDGES:
            nrng = np.random.default_rng(977)
            g2 = dict(g)
            for i in range(int(g['hparams'][5]) + 1):
                for k in ('kernel%d' % i, 'bias%d' % i):
                    g2[k] = (g[k] * (1.0 + nudge * nrng.uniform(-1, 1, g[k].shape))).astype(np.float32).astype(np.float64)
            if relu_tie_count(g2):
                continue
            tr2, _ = oracle_trainer(g2)
            _, _, gr2 = tr2.loss_and_grad(t(g['t_frames']), t(tg['target']), t(tg['sigma']), t(tg['offset']), scale, dt)
            gref2 = np.concatenate([np.concatenate([gr2[i].numpy().ravel(), gr2[n + i].numpy().ravel()]) for i in range(n)])
            p2 = pred.engine().flatten(golden_tree(g2)).requires_grad_(True)
            tree2 = network.ParamTree(); tree2.flat = p2
            loss2, _ = network.loss_fn_image(tree2, pred.apply, tg['target'], tg['sigma'], tg['offset'], g['t_frames'], rt['coords'], rt['Omega'],
                                             rt['J'], rt['g'], rt['dtau'], rt['Sigma'], rt['t_start_obs'], rt['t_geos'], rt['t_injection'], scale, units.hr, dt)
            loss2.backward()
            gd2 = p2.grad.cpu().numpy().astype(np.float64)
            e2 = np.abs(gd2 - gref2).max() / np.abs(gref2).max()
            assert e2 < gtol and l2err(gd2, gref2) < l2tol, ('f32', err, ties, 'NOT a tie: nudged %g -> %g' % (nudge, e2))
            return
        raise AssertionError(('f32', err, ties, 'ties persist under every nudge'))
    assert err < gtol, (err, ties)
    assert l2err(gdev, gref) < l2tol, (l2err(gdev, gref), ties)
    if dt == 'full':      # finite differences of the reference's own loss_fn_image (float64)
        eng = pred.engine()
        for (li, i, j), fd in zip(g['fd_idx'], g['fd_val']):
            idx = eng.kernel_off[li] + i * eng.out_dim[li] + j if i >= 0 else eng.bias_off[li] + j
            assert abs(gdev[idx] - fd) <= gtol * np.abs(g['fd_val']).max() + 2e-5 * abs(fd)


@pytest.mark.parametrize('tag', ['a', 'b', 'f'])
def test_training_steps_match_oracle(dev, golden, tag):
    """gradient_step_image (pack -> render -> chi2 -> backward -> Adam) x3 == oracle CpuTrainer (f32 mode)."""
    from bhnerf_amd import network, units
    g = golden('g5_predict_' + tag)
    tg = targets(g, 'full')
    tr, t = oracle_trainer(g, num_iters=3, lr_init=1e-3, lr_final=1e-5)
    pred, rt = device_setup(g, 'f32', dev)
    state = pred.init_state(network.ParamTree(golden_tree(g)), num_iters=3, lr_init=1e-3, lr_final=1e-5)
    for step in range(3):
        lref, _ = tr.step(t(g['t_frames']), t(tg['target']), t(tg['sigma']), t(tg['offset']), 1.0, 'full')
        loss, state, images = network.gradient_step_image(
            state, units.hr, 'full', tg['target'], tg['sigma'], tg['offset'], g['t_frames'], rt['coords'], rt['Omega'],
            rt['J'], rt['g'], rt['dtau'], rt['Sigma'], rt['t_start_obs'], rt['t_geos'], rt['t_injection'], 1.0)
        assert loss.shape == (1,) and abs(loss.item() - lref.item()) < 1e-4 * abs(lref.item())
        assert images.shape[:2] == (1, len(g['t_frames']))
    assert state.step == 3
    pref = flat_from_lists(tr.k, tr.b)
    pdev = state.flat.cpu().numpy()
    p0 = pred.engine().flatten(golden_tree(g)).cpu().numpy()
    moved = np.abs(pref - p0).max()
    assert moved > 1e-4
    # Adam normalises the step, so parameters move ~lr per step; compare against the total movement
    assert np.abs(pdev - pref).max() < 2e-2 * moved


def test_workspace_frame_chunking_is_equivalent(dev, golden):
    """A workspace that only holds one frame of tape must give the same gradient (slab accumulation)."""
    from bhnerf_amd import engine
    g = golden('g5_predict_e')
    pred, rt = device_setup(g, 'f32', dev)
    eng = pred.engine()
    eng.pack(eng.flatten(golden_tree(g)))
    geom = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
    tM0 = engine.frame_offsets(g['t_frames'], 0.0, rt['t_injection'], onp.GM_C3_SGRA_HR, dev)
    dimg = torch.rand((len(g['t_frames']), 1, geom.R), device=dev)
    full = eng.render_bwd(geom, tM0, dimg).clone()
    eng._ws, eng.max_workspace_bytes = None, 1          # forces the one-frame minimum
    chunked = eng.render_bwd(geom, tM0, dimg).clone()
    assert eng._ws.numel() < 0.6 * 1e12
    assert torch.allclose(full, chunked, rtol=1e-5, atol=1e-6 * float(full.abs().max()))
    assert float(full.abs().max()) > 0


@pytest.mark.parametrize('dt', ['full', 'lc'])
def test_frame_group_taped_step_equals_full_step(dev, golden, dt):
    """When the tape of all frames does not fit the workspace cap, gradient_step_image runs frame group by
    frame group on the taped path (chi^2 is a sum of per-frame terms): same loss, images and parameters
    as the all-frames step (f32; gradients are summed in a different order, hence the tolerance)."""
    from bhnerf_amd import network, units, _hip
    import ctypes as C
    g = golden('g5_predict_e')
    tg = targets(g, dt)
    B = len(g['t_frames'])
    assert B >= 2
    res = []
    for cap_frames in (None, 1):
        pred, rt = device_setup(g, 'f32', dev)
        eng = pred.engine()
        if cap_frames is not None:      # cap = tape of exactly one frame
            geom = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
            eng.max_workspace_bytes = int(_hip.lib().bhn_render_bwd_workspace_bytes(C.byref(eng.model), eng.mode, cap_frames,
                                                                                      geom.P_eff, dev.index or 0))
            assert not eng.fits_tape(B, geom.P_eff) and eng.tape_group(B, geom.P_eff) == cap_frames
        state = pred.init_state(network.ParamTree(golden_tree(g)), num_iters=2, lr_init=1e-3, lr_final=1e-4)
        for _ in range(2):
            loss, state, images = network.gradient_step_image(
                state, units.hr, dt, tg['target'], tg['sigma'], tg['offset'], g['t_frames'], rt['coords'], rt['Omega'],
                rt['J'], rt['g'], rt['dtau'], rt['Sigma'], rt['t_start_obs'], rt['t_geos'], rt['t_injection'], 1.0)
        res.append((loss.item(), images.clone(), state.flat.clone()))
    (l0, i0, p0), (l1, i1, p1) = res
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    assert torch.allclose(i0, i1, rtol=1e-5, atol=1e-6 * float(i0.abs().max()))
    assert torch.allclose(p0, p1, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('width,depth,S,deg', [(256, 4, 0, 3), (128, 4, 3, 3), (64, 8, 2, 3), (32, 6, 0, 3),
                                               (128, 4, 0, 0), (256, 4, 3, 1), (64, 4, 0, 2), (128, 4, 2, 4),
                                               (100, 4, 0, 3), (48, 4, 3, 3), (200, 6, 0, 2), (20, 4, 0, 3),
                                               (64, 5, 0, 3), (128, 7, 2, 3), (32, 3, 0, 3), (64, 2, 3, 2), (256, 5, 0, 3),
                                               (256, 8, 0, 3), (256, 8, 3, 3), (64, 4, 1, 3), (256, 6, 1, 2)])      # 8x256 f32: the LDS budget of the training forward
def test_random_problem_f32_and_bf16(dev, width, depth, S, deg):
    """Larger ragged problem (G=50 rays straddle wave tiles, several workgroup tiles, pre-injection
    and out-of-domain samples) against the float64 oracle; positional-encoding degrees 0..4 (network.py:98-122,
    `NeRF_Predictor.posenc_deg`); hidden widths that are not 32/64/128/256 run zero-padded on the next kernel width
    (the flat parameters and gradients keep the model's width); odd depths (and depth 2) with do_skip feed the
    skip-concat into the OUTPUT layer (network.py:59-62)."""
    prob = random_problem(width, depth, S, deg)
    for mode in ('f32', 'bf16'):
        ierr, gerr, l2 = random_problem_errors(prob, mode, dev)
        tol_img = {'f32': 1e-5, 'bf16': 1e-2}[mode]
        assert ierr < tol_img, (mode, ierr)
        ties = prob['ties']
        if mode == 'f32' and ties and not (gerr < GTOL[mode] and l2 < L2TOL[mode]):
            adjudicate_relu_ties(width, depth, S, deg, dev, gerr, ties)        # raises unless the ties explain it
            continue
        assert gerr < GTOL[mode], (mode, gerr, ties)
        assert l2 < L2TOL[mode], (mode, l2, ties)


@pytest.mark.parametrize('width,depth,S,deg', [(128, 4, 3, 5), (512, 4, 0, 3), (300, 3, 1, 6), (64, 2, 0, 7), (384, 8, 2, 4),
                                               (512, 8, 0, 8), (40, 5, 3, 10)])
def test_shapes_outside_the_fused_kernels(dev, width, depth, S, deg):
    """posenc_deg 5..10 (network.py:98-122: 3 + 6 deg <= 63 encoded inputs) and net_width 257..512 (network.py:154) -- no
    reference driver sets either -- run on the general layer-by-layer path (csrc/general_mlp.hip): f32 MFMAs in the f32 mode,
    held to the f32 bounds (octave i multiplies the f32 rounding of the warped coordinate by 2^i before the sine: the image
    bound grows with the degree); round 6: bf16 MFMAs on fragment-ordered operands in the bf16 mode, held to the bf16 bounds
    of the fused kernels.  Both modes: bitwise reproducible run to run (tiles of eight groups whose ray segments are combined
    in LDS, one atomic per tile and ray; slabs summed in a fixed order)."""
    prob = random_problem(width, depth, S, deg)
    tol_f32 = 1e-5 * max(1.0, 2.0 ** (deg - 5))
    for mode in ('f32', 'bf16'):
        out = []
        ierr, gerr, l2 = random_problem_errors(prob, mode, dev, grad_out=out)
        print('general path %dx%d deg %d S %d %s: image %.2e  gradient max %.2e  L2 %.2e  ties %d' % (depth, width, deg, S, mode, ierr, gerr, l2, prob['ties']))
        again = []
        random_problem_errors(prob, mode, dev, grad_out=again)
        assert np.array_equal(out[0], again[0]), (mode, 'not reproducible', np.abs(out[0] - again[0]).max())
        if mode == 'bf16':
            assert ierr < 1e-2, (mode, ierr)
            assert gerr < GTOL['bf16'], (mode, gerr, prob['ties'])
            assert l2 < L2TOL['bf16'], (mode, l2, prob['ties'])
            continue
        assert ierr < tol_f32, (mode, ierr)
        if prob['ties'] and not (gerr < GTOL['f32'] * tol_f32 / 1e-5 and l2 < L2TOL['f32'] * tol_f32 / 1e-5):
            adjudicate_relu_ties(width, depth, S, deg, dev, gerr, prob['ties'])
            continue
        assert gerr < GTOL['f32'] * tol_f32 / 1e-5 and l2 < L2TOL['f32'] * tol_f32 / 1e-5, (mode, gerr, l2)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_general_path_emission_and_workspace_chunks(dev, mode):
    """The general path's predictor output against the float64 oracle, its gradient through the recompute entry point
    (`bhn_render_bwd`) against the training pair (`bhn_render_fwd_train` records the tape, `bhn_render_bwd_tape` runs the delta chain
    from it: round 6), a workspace of 16 groups of tape against one that holds all, and run-to-run bitwise reproducibility."""
    from bhnerf_amd import units, network, engine as E
    prob = random_problem(320, 4, 2, 6)
    g = prob['g']
    hp = g['hparams']
    pred, rt = device_setup(g, mode, dev)
    tree = golden_tree(g)
    e = pred.apply({'params': tree}, g['t_frames'], units.hr, rt['coords'], rt['Omega'], 0.0, rt['t_geos'], prob['t_inj']).cpu().numpy()
    ks, bs = ot.tree_to_lists(tree)
    t = lambda x: torch.tensor(x, dtype=torch.float64)
    e_ref = ot.predictor(ks, bs, t(g['t_frames']), t(g['coords']), t(g['Omega']), 0.0, t(g['t_geos']), prob['t_inj'], onp.GM_C3_SGRA_HR,
                         hp[0], hp[1], hp[2], hp[3], posenc_deg=6, net_depth=4).numpy().reshape(e.shape)
    assert (e != 0).any() and np.abs(e - e_ref).max() / np.abs(e_ref).max() < (2e-5 if mode == 'f32' else 2e-2)
    eng = pred.engine()
    geom = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], rt['J'], rt['g'], rt['dtau'], rt['Sigma'])
    B = len(g['t_frames'])
    tM0 = E.frame_offsets(g['t_frames'], 0.0, prob['t_inj'], onp.GM_C3_SGRA_HR, dev)
    flat = eng.flatten(tree)
    eng.pack(flat)
    dimg = torch.rand((B, geom.Sx, geom.R), device=dev) - 0.3
    img = eng.render(geom, tM0).clone()
    img_t = eng.render_train(geom, tM0).clone()
    assert torch.equal(img, img_t) and float(img.abs().max()) > 0
    g_tape = eng.render_bwd_tape(geom, tM0, dimg).clone()
    g_rec = eng.render_bwd(geom, tM0, dimg).clone()
    assert torch.equal(g_tape, g_rec) and float(g_rec.abs().max()) > 0
    # a workspace SMALLER than one frame of tape (the call used to insist on a whole frame -- 35 GB for config 2 at
    # 8x512): slabs + 16 groups (two tiles of eight), so the chunks cut the frames anywhere; less is refused with BHN_EWORKSPACE
    import ctypes as C
    from bhnerf_amd import _hip
    lib = _hip.lib()
    small = int(lib.bhn_render_bwd_workspace_bytes(C.byref(eng.model), eng.mode, 1, 16 * 32, 0))
    one = int(lib.bhn_render_bwd_workspace_bytes(C.byref(eng.model), eng.mode, 1, geom.P_eff, 0))
    assert 0 < small < one and (geom.P_eff + 31) // 32 > 16
    # ... through the engine: a predictor whose workspace cap is that size takes the recompute route in chunks
    pred2 = network.NeRF_Predictor(hp[0], hp[1], hp[2], hp[3], posenc_deg=6, net_depth=4, net_width=320, mode=mode, device=dev)
    eng2 = pred2.engine()
    eng2.max_workspace_bytes = small
    eng2.pack(flat)
    assert eng2.workspace(B, geom.P_eff).numel() == small < eng.workspace(B, geom.P_eff).numel() and not eng2.fits_tape(B, geom.P_eff)
    g_chunks = eng2.render_bwd(geom, tM0, dimg)
    assert torch.allclose(g_chunks, g_rec, rtol=1e-5, atol=1e-6 * float(g_rec.abs().max()))
    # the recorded-tape pair refuses a workspace that cannot hold the whole tape (as the fused paths do)
    with pytest.raises(_hip.HipError, match='workspace'):
        eng2.render_train(geom, tM0); eng2.render_bwd_tape(geom, tM0, dimg)
    # run to run: images and gradients of the general path are bitwise reproducible (8-group tiles, combined ray sums, slabs)
    assert torch.equal(eng.render(geom, tM0), img) and torch.equal(eng.render_train(geom, tM0), img)
    assert torch.equal(eng.render_bwd_tape(geom, tM0, dimg), g_tape)
    ws = torch.empty((small,), dtype=torch.uint8, device=dev)
    out = torch.zeros_like(g_rec)
    gs, fs = geom.c_struct_fused(), eng._frames(tM0)
    call = lambda nbytes: lib.bhn_render_bwd(C.byref(eng.model), eng.mode, _hip.ptr(eng.packed), C.byref(gs), C.byref(fs), _hip.ptr(dimg),
                                             _hip.ptr(out), _hip.ptr(ws), nbytes, _hip.stream_ptr(dev))
    _hip.check(call(small))
    assert torch.allclose(out, g_rec, rtol=1e-5, atol=1e-6 * float(g_rec.abs().max()))
    assert call(small - 4096) == 4 and b'workspace' in lib.bhn_last_error()          # BHN_EWORKSPACE


@pytest.mark.parametrize('rows', [48, 47])
def test_reference_default_network_on_twelve_and_eight_group_tiles(dev, rows):
    """The fused 4x128 path runs its two forward kernels on 12-wave workgroups (tiles of 12 point groups, PolBF16X) when the ray set has
    >= 3072 groups per frame and on 8-wave ones below: 48 x 32 rays x 64 samples = 3072 groups, 47 x 32 = 3008.  On both sides of the
    threshold: images and the chi^2 gradient against the f64 oracle on the WHOLE problem, `render` == `render_train` bit for bit, the
    recorded-tape gradient == the recompute route bit for bit (the fused backward walks either tape in quads of groups)."""
    from bhnerf_amd import network, synthetic, units, engine as E
    H, Wd, G, B = rows, 32, 64, 2
    geo = synthetic.synthetic_geodesics(H, Wd, G, fov_M=16.0, inc_deg=60.0, seed=7)
    dom = (8.0, 0.0, np.inf, np.inf)
    rng = np.random.default_rng(21)
    tree = onp.he_uniform_params(rng, 4, 128, 21, dtype=np.float32)
    for i in range(5):
        tree['MLP_0']['Dense_%d' % i]['bias'] = rng.uniform(-0.05, 0.05, tree['MLP_0']['Dense_%d' % i]['bias'].shape).astype(np.float32)
    tree['MLP_0']['Dense_4']['bias'] = tree['MLP_0']['Dense_4']['bias'] + 9.0
    t_frames = np.array([0.1, 0.6])
    target = rng.uniform(0, 1e-2, (B, H, Wd)); sigma = rng.uniform(0.5, 2.0, (B, H, Wd)); offset = np.zeros((B, H, Wd))
    t64 = lambda x: torch.tensor(np.asarray(x, dtype=np.float64))
    f32r = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
    ks, bs = ot.tree_to_lists(tree, torch.float64)
    geom_t = dict(coords=t64(f32r(geo['coords'])), Omega=t64(f32r(geo['Omega'])), t_geos=t64(f32r(geo['t_geos'])), g=t64(f32r(geo['g'])),
                  dtau=t64(f32r(geo['dtau'])), Sigma=t64(f32r(geo['Sigma'])), J=None, t_start_obs=0.0, t_injection=float(geo['t_injection']))
    hp = dict(GM_c3=onp.GM_C3_SGRA_HR, scale=dom[0], rmin=dom[1], rmax=dom[2], z_width=dom[3], posenc_deg=3, net_depth=4)
    tr = ot.CpuTrainer(ks, bs, geom_t, hp)
    loss_ref, img_ref, grads_ref = tr.loss_and_grad(t64(t_frames), t64(target), t64(sigma), t64(offset), 1.0, 'full')
    n = len(tr.k)
    gref = np.concatenate([np.concatenate([grads_ref[i].numpy().ravel(), grads_ref[n + i].numpy().ravel()]) for i in range(n)])
    pred = network.NeRF_Predictor(*dom, net_depth=4, net_width=128, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    assert geom.compact is None and geom.P_eff // 32 == rows * 64                       # dense layout: rows * 32 * 64 / 32 groups per frame
    params = eng.flatten(tree).requires_grad_(True)
    ptree = network.ParamTree(); ptree.flat = params
    loss, [images] = network.loss_fn_image(ptree, pred.apply, target, sigma, offset, t_frames, geo['coords'], geo['Omega'], 1.0, geo['g'],
                                           geo['dtau'], geo['Sigma'], 0.0, geo['t_geos'], float(geo['t_injection']), 1.0, units.hr, 'full')
    loss.backward()
    img = images.detach().cpu().numpy().reshape(img_ref.shape)
    assert np.abs(img - img_ref.numpy()).max() < 1e-2 * float(img_ref.abs().max())
    gdev = params.grad.cpu().numpy().astype(np.float64)
    err = float(np.linalg.norm(gdev - gref) / np.linalg.norm(gref))
    print('4x128 bf16, %d x 32 rays (%d groups per frame): gradient vs f64 oracle rel L2 %.3e, max-norm %.3e' % (rows, rows * 64, err, float(np.abs(gdev - gref).max() / np.abs(gref).max())))
    assert err < 3e-2 and abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item())
    tM0 = E.frame_offsets(t_frames, 0.0, float(geo['t_injection']), onp.GM_C3_SGRA_HR, dev)
    eng.pack(eng.flatten(tree))
    a = eng.render(geom, tM0).clone(); b = eng.render_train(geom, tM0).clone()
    assert torch.equal(a, b) and float(a.abs().max()) > 0
    dimg = torch.rand((B, 1, geom.R), device=dev) - 0.4
    g_tape = eng.render_bwd_tape(geom, tM0, dimg).clone()
    g_rec = eng.render_bwd(geom, tM0, dimg).clone()
    assert torch.equal(g_tape, g_rec) and float(g_tape.abs().max()) > 0


def adjudicate_relu_ties(width, depth, S, deg, dev, gerr, ties):
    """A detected ReLU tie is only an explanation if the SAME problem without the tied ray samples meets the f32 bounds:
    those samples get Doppler weight g = 0 on both sides (no contribution to the image, hence none to the gradient), every
    other sample keeps its inputs and its forward values.  Anything else is reported as a failure of the original."""
    prob = random_problem(width, depth, S, deg, drop_ties=True)
    assert prob['ties_left'] == 0 and 0 < prob['dropped'] <= ties
    ierr, g2, l2 = random_problem_errors(prob, 'f32', dev)
    assert ierr < 1e-5 and g2 < GTOL['f32'] and l2 < L2TOL['f32'], ('f32', gerr, ties, 'NOT a tie: without the %d tied samples %g / %g' % (prob['dropped'], g2, l2))
    print('relu ties adjudicated: %d ties on %d ray samples, gradient error %.2e; without those samples: error %.2e (L2 %.2e)'
          % (ties, prob['dropped'], gerr, g2, l2))


def random_problem(width, depth, S, deg, nudge=0.0, drop_ties=False):
    rng = np.random.default_rng(width + depth)
    H, Wd, G, B = RANDOM_PROBLEM_SHAPE
    ja, jb, js = RANDOM_PROBLEM_JITTER
    alpha, beta = np.meshgrid(np.linspace(-8 + ja, 8 + ja, H), np.linspace(-8 + jb, 8 + jb, Wd), indexing='ij')
    s = np.linspace(-9.6 + js, 9.6 + js, G)
    inc = np.deg2rad(60.0)
    coords = np.stack([alpha[..., None] * np.ones(G), beta[..., None] * np.cos(inc) + s * np.sin(inc),
                       -beta[..., None] * np.sin(inc) + s * np.cos(inc)])
    r = np.sqrt((coords ** 2).sum(0)) + 0.3
    geo = dict(coords=coords, Omega=1.0 / (r ** 1.5 + 0.1), t_geos=-(1000.0 - (s + 9.6)) * np.ones_like(r),
               g=rng.uniform(0.6, 1.4, r.shape), Sigma=r ** 2, dtau=(s[1] - s[0]) / r ** 2)
    J = None
    if S:
        I = rng.uniform(0.5, 1.5, r.shape); chi = rng.uniform(0, np.pi, r.shape)
        J = np.stack([I, 0.85 * I * np.cos(2 * chi), 0.85 * I * np.sin(2 * chi)])[:S]
    t_frames = np.sort(rng.uniform(0, 1, B)); t_inj = -(1000.0 - 4.0)
    # identical inputs on both sides: everything is rounded to float32 first, the oracle then
    # evaluates those values in float64
    f32r = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
    geo = {k: f32r(v) for k, v in geo.items()}
    J = f32r(J) if S else None
    tree = onp.he_uniform_params(rng, depth, width, 3 + 6 * deg, dtype=np.float32)
    nrng = np.random.default_rng(977)
    for i in range(depth + 1):
        d = tree['MLP_0']['Dense_%d' % i]
        d['kernel'] = d['kernel'].astype(np.float64)
        d['bias'] = f32r(rng.uniform(-0.1, 0.1, d['bias'].shape))
        if nudge:                            # tie adjudication: every weight moved by a random relative amount <= nudge
            d['kernel'] = f32r(d['kernel'] * (1.0 + nudge * nrng.uniform(-1, 1, d['kernel'].shape)))
            d['bias'] = f32r(d['bias'] * (1.0 + nudge * nrng.uniform(-1, 1, d['bias'].shape)))
    g = dict(geo, J=(J if S else np.array(1.0)), t_frames=t_frames, t_start_obs=0.0, t_injection=t_inj,
             hparams=np.array(list(RANDOM_PROBLEM_DOMAIN) + [deg, depth, width, 1.0]))
    for i in range(depth + 1):
        g['kernel%d' % i] = tree['MLP_0']['Dense_%d' % i]['kernel']; g['bias%d' % i] = tree['MLP_0']['Dense_%d' % i]['bias']
    ties, tied = relu_tie_count(g, return_points=True)
    dropped, ties_left = 0, ties
    if drop_ties:                            # tie adjudication: the tied ray samples no longer reach the image
        g['g'] = np.where(tied, 0.0, g['g'])
        dropped = int(tied.sum())
        # re-run the detection on the MODIFIED problem: tied samples that still reach the image (Doppler weight != 0)
        _, tied2 = relu_tie_count(g, return_points=True)
        ties_left = int((tied2 & (np.broadcast_to(g['g'], tied2.shape) != 0)).sum())
    tr, t = oracle_trainer(g)
    shape = (B, S, H, Wd) if S else (B, H, Wd)
    target = rng.uniform(0, 1e-3, shape); sigma = rng.uniform(0.5, 2.0, shape); offset = np.zeros(shape)
    loss_ref, img_ref, grads_ref = tr.loss_and_grad(t(t_frames), t(target), t(sigma), t(offset), 1.0, 'full')
    n = len(tr.k)
    gref = np.concatenate([np.concatenate([grads_ref[i].numpy().ravel(), grads_ref[n + i].numpy().ravel()]) for i in range(n)])
    return dict(g=g, S=S, t_frames=t_frames, t_inj=t_inj, target=target, sigma=sigma, offset=offset, img_ref=img_ref, gref=gref,
                ties=ties, dropped=dropped, ties_left=ties_left)


def random_problem_errors(prob, mode, dev, grad_out=None):
    """(image error / image maximum, gradient max error / largest entry, gradient relative L2) of the HIP path against the
    float64 oracle on a random_problem(); `grad_out` (a list) receives the device gradient."""
    from bhnerf_amd import network, units
    g, S, img_ref, gref = prob['g'], prob['S'], prob['img_ref'], prob['gref']
    pred, rt = device_setup(g, mode, dev)
    params = pred.engine().flatten(golden_tree(g)).requires_grad_(True)
    ptree = network.ParamTree(); ptree.flat = params
    # one Stokes plane: the reference squeezes the unit axis (network.py:418), so its images -- and the targets a caller
    # pairs them with -- are (B, H, W); the oracle trainer keeps the axis
    sq = (lambda v: v[:, 0]) if S == 1 else (lambda v: v)
    loss, [images] = network.loss_fn_image(ptree, pred.apply, sq(prob['target']), sq(prob['sigma']), sq(prob['offset']), prob['t_frames'],
                                           rt['coords'], rt['Omega'], rt['J'], rt['g'], rt['dtau'], rt['Sigma'], 0.0, rt['t_geos'],
                                           prob['t_inj'], 1.0, units.hr, 'full')
    loss.backward()
    img = images.detach().cpu().numpy().reshape(img_ref.shape)
    gdev = params.grad.cpu().numpy()
    if float(img_ref.abs().max()) == 0.0 or float(np.abs(gref).max()) == 0.0:
        # an empty recovery domain (the fuzz draws thin shells no sample falls into): the reference image and gradient are
        # identically zero, and so must the device's be -- there is no scale to take an error relative to
        assert float(img_ref.abs().max()) == 0.0 and float(np.abs(gref).max()) == 0.0
        assert not img.any() and not gdev.any(), (mode, 'non-zero output for an empty domain')
        return 0.0, 0.0, 0.0
    ierr = np.abs(img - img_ref.numpy()).max() / img_ref.abs().max().item()
    if grad_out is not None:
        grad_out.append(gdev)
    return ierr, np.abs(gdev - gref).max() / np.abs(gref).max(), l2err(gdev, gref)


@pytest.mark.parametrize('depth,S,deg', [(4, 0, 3), (4, 3, 1), (6, 1, 2), (8, 0, 3)])
def test_tape8_mode_gradient(dev, depth, S, deg):
    """BHN_BF16_T8 (include/bhnerf_hip.h: bf16 arithmetic, the backward's tape of h_l / gA_l as e4m3 bytes; width 256): same
    images as the bf16 mode, a gradient inside the bf16 mode's bounds against the float64 oracle and close to the bf16 mode's own
    gradient.  The rounding of the 8-bit operands is ~3 % per element and averages out over the points of the sum: 4e-2 of the
    gradient's norm on this problem of a few thousand points in the domain; 2e-3 from 30 k points on
    (test_tape8_mode_scales_follow_the_gradient, tools/dbg_t8.py; tools/exp_fp8_tape_accuracy.py predicts 2.4e-3)."""
    prob = random_problem(256, depth, S, deg)
    g16, g8 = [], []
    i16, _, _ = random_problem_errors(prob, 'bf16', dev, g16)
    i8, gerr, l2 = random_problem_errors(prob, 'bf16_t8', dev, g8)
    assert i8 == i16                                           # the forward is the bf16 mode's
    assert gerr < GTOL['bf16'] and l2 < L2TOL['bf16'], (gerr, l2)
    assert l2err(g8[0], g16[0]) < 4e-2, l2err(g8[0], g16[0])


def test_tape8_mode_scales_follow_the_gradient(dev):
    """The 8-bit tape stores gA_l with one power-of-two scale per layer: the ratio |gA_l|max / |dimages|max of the PREVIOUS
    backward call on the workspace (the first call calibrates itself) times |dimages|max of this call, with 16x head room.
    Jumps of the loss scale in either direction are therefore exact at once; ratios that are off by more than the head room
    (stale state, weights replaced wholesale without BHN_T8_CALIBRATE) make the tape values LIMITED -- finite, no NaN -- and
    recover by up to 16x per call."""
    from bhnerf_amd import engine, network, synthetic, constants
    geo = synthetic.synthetic_geodesics(16, 16, 64, seed=5)
    out = {}
    for mode in ('bf16', 'bf16_t8'):
        pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=256, mode=mode, device=dev)
        eng = pred.engine()
        geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
        flat = eng.flatten(network.MLP(4, 256).init(1, 21))
        eng.pack(flat)
        tM0 = engine.frame_offsets(np.linspace(0, 0.8, 2), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
        gen = torch.Generator(device='cpu').manual_seed(7)
        dimg = ((torch.rand((2, 1, geom.R), generator=gen) - 0.3) * 1e-3).to(dev)
        one_frame = dimg.clone(); one_frame[1] = 0.0                       # the residuals somewhere else: the ratios move a little
        res = []
        for d, factor in ((dimg, 1.0), (dimg, 8.0), (dimg, 1e5), (dimg, 1e-4), (one_frame, 1.0), (dimg, 1.0)):
            eng.render_train(geom, tM0)
            res.append(eng.render_bwd_tape(geom, tM0, factor * d).cpu().numpy() / factor)
        if mode == 'bf16_t8':
            # stale state: the ratios 1000x too small (layout of the state block: fused_bwd.hip t8_prepare_kernel; it sits behind
            # the dW slabs, one per compute unit).  The tape values are limited to the range, the maxima seen are those of the
            # limited chain, so the ratios recover by at most the head room (16x) per call
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            off = ncu * 9 * 10 * 1024 * 4
            st = eng._ws[off:off + 128].view(torch.float32)
            st[16:24] *= 1e-3
        for _ in range(4):
            eng.render_train(geom, tM0)
            res.append(eng.render_bwd_tape(geom, tM0, dimg).cpu().numpy())
        out[mode] = res
    errs = [l2err(a, b) for a, b in zip(out['bf16_t8'], out['bf16'])]
    assert all(np.isfinite(a).all() for a in out['bf16_t8']), errs
    assert max(errs[:6]) < 5e-3, errs
    assert errs[6] > 5e-3 and errs[9] < 5e-3, errs      # limited right after the state went stale, accurate again three calls later


def test_tape8_mode_rejects_other_networks(dev):
    """The 8-bit tape kernels exist for width 256, depth >= 3, no skip-concat into layer 1 or the output layer: anything else
    is BHN_EUNSUPPORTED (loud), never a silent fallback.  Forward-only entry points treat the mode as bf16."""
    from bhnerf_amd import engine, network, synthetic, constants, _hip
    geo = synthetic.synthetic_geodesics(8, 8, 32, seed=5)
    for width, depth in ((128, 4), (256, 2), (256, 5)):
        pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=depth, net_width=width, mode='bf16_t8', device=dev)
        eng = pred.engine()
        geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
        eng.pack(eng.flatten(network.MLP(depth, width).init(1, 21)))
        tM0 = engine.frame_offsets(np.linspace(0, 0.8, 2), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
        assert torch.isfinite(eng.render(geom, tM0)).all()
        with pytest.raises(_hip.HipError):
            eng.render_train(geom, tM0)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('tag', ['b', 'c', 'f'])
def test_taped_training_path_equals_recompute_path(dev, golden, tag, mode):
    """bhn_render_fwd_train + bhn_render_bwd_tape (forward recorded on the tape, delta chain only) must give
    the images of bhn_render_fwd and the gradient of bhn_render_bwd (recompute) -- same arithmetic."""
    from bhnerf_amd import engine
    g = golden('g5_predict_' + tag)
    pred, rt = device_setup(g, mode, dev)
    eng = pred.engine()
    eng.pack(eng.flatten(golden_tree(g)))
    geom = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], rt['J'] if g['J'].ndim else None, rt['g'], rt['dtau'], rt['Sigma'])
    tM0 = engine.frame_offsets(g['t_frames'], 0.0, rt['t_injection'], onp.GM_C3_SGRA_HR, dev)
    dimg = torch.randn((len(g['t_frames']), geom.Sx, geom.R), device=dev)
    img_ref = eng.render(geom, tM0).clone()
    grad_ref = eng.render_bwd(geom, tM0, dimg).clone()
    assert eng.fits_tape(int(tM0.numel()), geom.P_eff)
    img = eng.render_train(geom, tM0)
    grad = eng.render_bwd_tape(geom, tM0, dimg)
    assert torch.allclose(img, img_ref, rtol=1e-6, atol=1e-7 * float(img_ref.abs().max()))
    assert torch.equal(grad, grad_ref)


@pytest.mark.parametrize('depth,width,B,do_skip', [(2, 64, 2, False), (3, 64, 2, False), (5, 64, 2, False),
                                                   (4, 128, 2, False),      # the fused width-128 backward with the skip tile discarded
                                                   (4, 128, 6, True),       # ... and at the reference's own batch size 6
                                                   (4, 128, 6, False)])     #     (scripts/Fit_ALMA_LP_Apr11_SgrA_Flare.yaml:46)
def test_networks_without_skip_connection_and_batch_of_six(dev, depth, width, B, do_skip):
    """do_skip=False with depths outside the reference's 4/6/8 (two hidden layers: the layer-1 weight-gradient job
    that recomputes h_1 is also the last hidden layer) and at width 128 / depth 4, where the gradient comes out of
    bwd128_kernel<4, 3> with the skip layer's encoded-input tile discarded by reduce128_kernel;
    frame batches of 6, the reference's own batch size: images and gradient vs the f64 oracle.  (Depth 1 is rejected.)"""
    from bhnerf_amd import network, units, _hip
    with pytest.raises(_hip.HipError):
        network.NeRF_Predictor(8.0, net_depth=1, net_width=64, do_skip=False, device=dev).engine()
    rng = np.random.default_rng(100 + depth + width + B)
    H, Wd, G = 6, 5, 40
    alpha, beta = np.meshgrid(np.linspace(-7, 7, H), np.linspace(-7, 7, Wd), indexing='ij')
    s = np.linspace(-9.0, 9.0, G)
    coords = np.stack([alpha[..., None] * np.ones(G), beta[..., None] * 0.5 + s * 0.8, -beta[..., None] * 0.8 + s * 0.5])
    r = np.sqrt((coords ** 2).sum(0)) + 0.3
    f32r = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
    geo = {k: f32r(v) for k, v in dict(coords=coords, Omega=1.0 / (r ** 1.5 + 0.1), t_geos=-(1000.0 - (s + 9.0)) * np.ones_like(r),
                                        g=rng.uniform(0.6, 1.4, r.shape), Sigma=r ** 2, dtau=(s[1] - s[0]) / r ** 2).items()}
    t_frames = np.linspace(0.1, 0.7, B); t_inj = -(1000.0 - 3.0)
    tree = onp.he_uniform_params(rng, depth, width, 21, do_skip=do_skip, dtype=np.float32)
    for i in range(depth + 1):
        d = tree['MLP_0']['Dense_%d' % i]
        d['kernel'] = d['kernel'].astype(np.float64); d['bias'] = f32r(rng.uniform(-0.1, 0.1, d['bias'].shape))
    tree['MLP_0']['Dense_%d' % depth]['bias'] = tree['MLP_0']['Dense_%d' % depth]['bias'] + 9.0
    t64 = lambda x: torch.tensor(np.asarray(x, dtype=np.float64))
    ks, bs = ot.tree_to_lists(tree, torch.float64)
    geom_t = dict(coords=t64(geo['coords']), Omega=t64(geo['Omega']), t_geos=t64(geo['t_geos']), g=t64(geo['g']), dtau=t64(geo['dtau']),
                  Sigma=t64(geo['Sigma']), J=None, t_start_obs=0.0, t_injection=t_inj)
    hp = dict(GM_c3=onp.GM_C3_SGRA_HR, scale=8.0, rmin=2.0, rmax=8.0, z_width=4.0, posenc_deg=3, net_depth=depth, do_skip=do_skip)
    tr = ot.CpuTrainer(ks, bs, geom_t, hp)
    target = rng.uniform(0, 1e-2, (B, H, Wd)); sigma = rng.uniform(0.5, 2.0, (B, H, Wd)); offset = np.zeros((B, H, Wd))
    loss_ref, img_ref, grads_ref = tr.loss_and_grad(t64(t_frames), t64(target), t64(sigma), t64(offset), 1.0, 'full')
    n = len(tr.k)
    gref = np.concatenate([np.concatenate([grads_ref[i].numpy().ravel(), grads_ref[n + i].numpy().ravel()]) for i in range(n)])
    assert np.abs(gref).max() > 0
    f = lambda k: np.ascontiguousarray(geo[k].astype(np.float32))
    for mode in ('f32', 'bf16'):
        pred = network.NeRF_Predictor(8.0, 2.0, 8.0, 4.0, net_depth=depth, net_width=width, do_skip=do_skip, mode=mode, device=dev)
        params = pred.engine().flatten(tree).requires_grad_(True)
        ptree = network.ParamTree(); ptree.flat = params
        loss, [images] = network.loss_fn_image(ptree, pred.apply, target, sigma, offset, t_frames, f('coords'), f('Omega'), 1.0, f('g'),
                                               f('dtau'), f('Sigma'), 0.0, f('t_geos'), t_inj, 1.0, units.hr, 'full')
        loss.backward()
        ierr = np.abs(images.detach().cpu().numpy().reshape(img_ref.shape) - img_ref.numpy()).max() / img_ref.abs().max().item()
        assert ierr < {'f32': 1e-5, 'bf16': 3e-2}[mode], (mode, ierr)
        assert l2err(params.grad.cpu().numpy(), gref) < L2TOL[mode], (mode, l2err(params.grad.cpu().numpy(), gref))
        # the taped route of the same gradient
        eng = pred.engine()
        geom = pred.geometry(f('coords'), f('Omega'), f('t_geos'), None, f('g'), f('dtau'), f('Sigma'))
        tM0, _ = network._frame_offsets(t_frames, units.hr, 0.0, t_inj, dev)
        eng.pack(params.detach())
        dimg = torch.rand((B, 1, geom.R), device=dev)
        eng.render_train(geom, tM0)
        assert torch.equal(eng.render_bwd_tape(geom, tM0, dimg), eng.render_bwd(geom, tM0, dimg))
