#!/opt/conda/bin/python3.9
"""Generate golden input/output vectors by running the REFERENCE's own code.

Run in the build container only (the reference never travels to the GPU box):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

How: /root/reference/bhnerf/{utils,constants,kgeo,emission,network}.py are loaded
unmodified through importlib with sys.modules pre-seeded so that ``jax.numpy`` IS
numpy (float64), ``xarray``/external ``kgeo`` are empty, and ``flax.linen`` is a
~40-line stand-in whose ``Module`` is a dataclass base and whose ``Dense`` is
``x @ kernel + bias`` on injected weights (flax 0.3.4 semantics, kernel (in,out)).
With that, the reference's *own* ``velocity_warp_coords``, ``fill_unsupervised_emission``,
``radiative_trasfer``, ``posenc``, ``MLP.__call__``, ``NeRF_Predictor.__call__``,
``image_plane_prediction``, ``loss_fn_image`` and ``loss_fn_eht`` execute, and their
inputs/outputs are written to ``tests/golden/*.npz``.  Only data is committed.

What is NOT pinned by this: flax's real Dense (restated), jax PRNG init, optax Adam,
jax.grad.  Gradients are pinned by central finite differences of the reference's own
``loss_fn_image`` (G9).
"""
import dataclasses
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

warnings.filterwarnings('ignore')
REF = '/root/reference/bhnerf'
OUT = os.path.dirname(os.path.abspath(__file__))

# ---- numpy-compat shims astropy 4.3 needs on numpy 1.26 -------------------------------
for name, val in dict(asscalar=lambda a: a.item(), alen=len, msort=lambda a: np.sort(a, 0),
                      sometrue=np.any, alltrue=np.all, float=float, int=int, bool=bool,
                      object=object, complex=complex).items():
    if not hasattr(np, name):
        setattr(np, name, val)

# ---- stubs -----------------------------------------------------------------------------
_dense_params = []          # injected [(kernel, bias), ...] consumed in call order


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


jnp = np
jax = _mod('jax', numpy=np, jit=lambda f=None, **kw: (f if f is not None else (lambda g: g)),
           tree_map=lambda f, x: f(x), local_device_count=lambda: 1, device_count=lambda: 1)
sys.modules['jax.numpy'] = np
jax.nn = _mod('jax.nn', initializers=types.SimpleNamespace(he_uniform=lambda: 'he_uniform'))
jax.random = _mod('jax.random', PRNGKey=lambda s: s)
jax.lax = _mod('jax.lax')
import scipy.ndimage as _ndi  # noqa: E402
# jax.scipy.ndimage.map_coordinates(input, coordinates, order, mode='constant', cval=0.0): SciPy's own function is the
# algorithm JAX re-implements (order <= 1 only)
jax.scipy = _mod('jax.scipy', ndimage=types.SimpleNamespace(
    map_coordinates=lambda inp, coords, order, mode='constant', cval=0.0: _ndi.map_coordinates(inp, coords, order=order, mode=mode, cval=cval)))
sys.modules['jax.scipy.ndimage'] = jax.scipy.ndimage


_grid_param = []            # injected value of GRID_Predictor's `grid` parameter


class _Module:
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        dataclasses.dataclass(cls)

    def param(self, name, init_fn, *init_args):      # flax: the stored parameter (here: the injected array)
        value = _grid_param[0]
        assert tuple(value.shape) == tuple(init_args[0]), (value.shape, init_args)
        return value


class _Dense:
    def __init__(self, features, kernel_init=None):
        self.features = features

    def __call__(self, x):
        kernel, bias = _dense_params.pop(0)
        assert kernel.shape == (x.shape[-1], self.features), (kernel.shape, x.shape, self.features)
        return x @ kernel + bias


linen = _mod('flax.linen', Module=_Module, compact=lambda f: f, Dense=_Dense,
             relu=lambda x: np.maximum(x, 0), sigmoid=lambda x: 1.0 / (1.0 + np.exp(-x)))
flax = _mod('flax', linen=linen)
_mod('flax.training', train_state=types.SimpleNamespace(), checkpoints=types.SimpleNamespace())
_mod('flax.training.train_state')
_mod('flax.training.checkpoints')
_mod('optax')
_mod('xarray')
_mod('kgeo', __all__=[])
_mod('tqdm'); _mod('tqdm.auto', tqdm=lambda x, **k: x); _mod('tqdm.contrib', tzip=lambda *a, **k: zip(*a))
_mod('tensorboardX', SummaryWriter=object)

pkg = types.ModuleType('bhnerf')
pkg.__path__ = [REF]
sys.modules['bhnerf'] = pkg


def _load(name):
    spec = importlib.util.spec_from_file_location('bhnerf.' + name, os.path.join(REF, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules['bhnerf.' + name] = mod
    setattr(pkg, name, mod)
    spec.loader.exec_module(mod)
    return mod


constants = _load('constants')
utils = _load('utils')
kgeo = _load('kgeo')
emission = _load('emission')
network = _load('network')
from astropy import units  # noqa: E402

GM_c3_hr = float(constants.GM_c3(constants.sgra_mass).to('hr').value)


def mlp_dims(depth, width, fin):
    dims, cur, skip = [], fin, depth // 2
    for i in range(depth):
        dims.append((cur, width)); cur = width
        if i % skip == 0 and i > 0:
            cur = width + fin
    dims.append((cur, 1))
    return dims


def random_weights(rng, depth, width, fin):
    ws = []
    for fi, fo in mlp_dims(depth, width, fin):
        lim = np.sqrt(6.0 / fi)
        ws.append((rng.uniform(-lim, lim, (fi, fo)), rng.uniform(-0.1, 0.1, (fo,))))
    return ws


def geometry(rng, H, W, G, S=None, rmax=8.0):
    """Small synthetic 'geodesic' arrays (straight rays through the volume)."""
    alpha, beta = np.meshgrid(np.linspace(-rmax, rmax, H), np.linspace(-rmax, rmax, W), indexing='ij')
    s = np.linspace(-1.2 * rmax, 1.2 * rmax, G)
    inc = np.deg2rad(60.0)
    x = alpha[..., None] * np.ones(G)
    y = beta[..., None] * np.cos(inc) + s * np.sin(inc)
    z = -beta[..., None] * np.sin(inc) + s * np.cos(inc)
    coords = np.stack([x, y, z])
    r = np.sqrt((coords ** 2).sum(0)) + 0.3
    Omega = 1.0 / (r ** 1.5 + 0.1)
    t_geos = -(1000.0 - (s + 1.2 * rmax)) * np.ones_like(x)
    g = rng.uniform(0.6, 1.4, x.shape)
    Sigma = r ** 2
    dtau = (s[1] - s[0]) / Sigma
    J = None
    if S:
        I = rng.uniform(0.5, 1.5, x.shape); chi = rng.uniform(0, np.pi, x.shape)
        J = np.stack([I, 0.85 * I * np.cos(2 * chi), 0.85 * I * np.sin(2 * chi)])[:S]
    return dict(coords=coords, Omega=Omega, t_geos=t_geos, g=g, dtau=dtau, Sigma=Sigma, J=J)


ONLY = set(sys.argv[1:])       # optional: names of the fixtures to (re)write; default all


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print('wrote', name, {k: np.shape(v) for k, v in arrays.items()})


def main():
    rng = np.random.default_rng(20240501)

    # G0: constants ------------------------------------------------------------------
    save('g0_constants', GM_c3_hr=GM_c3_hr, isco0=constants.isco_pro(0.0), isco94=constants.isco_pro(0.94))

    # G1: velocity_warp_coords --------------------------------------------------------
    geo = geometry(rng, 4, 4, 5)
    t_frames = np.array([0.0, 0.31, 0.77])
    t_inj = -(1000.0 - 9.0)       # some samples are seen before injection -> NaN
    w_units = emission.velocity_warp_coords(geo['coords'], geo['Omega'], t_frames, 0.0, geo['t_geos'],
                                            t_inj, t_units=units.hr, use_jax=True)
    w_nounits = emission.velocity_warp_coords(geo['coords'], geo['Omega'], t_frames, 0.1, geo['t_geos'],
                                              t_inj, t_units=None, use_jax=False)
    w_scalarO = emission.velocity_warp_coords(geo['coords'], 0.05, t_frames, 0.0, geo['t_geos'],
                                              t_inj, t_units=units.hr, use_jax=True)
    w_scalart = emission.velocity_warp_coords(geo['coords'], geo['Omega'], 0.4, 0.0, 0.0, 0.0,
                                              t_units=None, use_jax=False)
    rot = utils.rotation_matrix([0, 0, 1], np.array([0.3, -1.2]))
    save('g1_warp', coords=geo['coords'], Omega=geo['Omega'], t_geos=geo['t_geos'], t_frames=t_frames,
         t_injection=t_inj, out_units=w_units, out_nounits=w_nounits, out_scalar_omega=w_scalarO,
         out_scalar_t=w_scalart, rot_angles=np.array([0.3, -1.2]), rot=rot,
         nan_fraction=np.isnan(w_units).mean())

    # G2: posenc ----------------------------------------------------------------------
    x = rng.uniform(-1.3, 1.3, (7, 3)); x[0] = [0.1, -0.2, 0.3]
    save('g2_posenc', x=x, **{'deg%d' % d: network.posenc(x, d) for d in (0, 1, 3, 5)})

    # G3: fill_unsupervised_emission ---------------------------------------------------
    geo3 = geometry(rng, 6, 5, 11, rmax=9.0)
    e3 = rng.uniform(0.1, 1.0, (2,) + geo3['Omega'].shape)
    f3 = emission.fill_unsupervised_emission(e3, geo3['coords'], rmin=5.0, rmax=8.0, z_width=4.0, use_jax=True)
    r2 = (geo3['coords'] ** 2).sum(0)
    save('g3_fill', emission=e3, coords=geo3['coords'], rmin=5.0, rmax=8.0, z_width=4.0, out=f3,
         hits=np.array([(r2 < 25).sum(), (r2 > 64).sum(), (np.abs(geo3['coords'][2]) > 4).sum()]))

    # G4: radiative_trasfer -------------------------------------------------------------
    geo4 = geometry(rng, 3, 4, 6)
    e4 = rng.uniform(0, 1, (2, 3) + geo4['Omega'].shape)
    save('g4_rt', emission=e4, g=geo4['g'], dtau=geo4['dtau'], Sigma=geo4['Sigma'],
         out_arrays=kgeo.radiative_trasfer(e4, geo4['g'], geo4['dtau'], geo4['Sigma'], use_jax=True),
         out_scalars=kgeo.radiative_trasfer(e4[0, 0], 1.3, 1.0, 0.5),
         out_3d=kgeo.radiative_trasfer(e4[1, 2], geo4['g'], geo4['dtau'], geo4['Sigma']))

    # G5/G6/G8: full predictor + image_plane_prediction + loss_fn_image -------------------
    for tag, (H, W, G, S, depth, width, nt) in {
            'a': (4, 4, 8, None, 4, 32, 3),      # unpolarised
            'b': (3, 4, 6, 3, 4, 64, 3),         # Stokes I,Q,U
            'c': (4, 3, 8, 2, 8, 32, 2),         # depth 8 (skip after layer 4), 2 Stokes
            'd': (4, 4, 8, 3, 4, 32, 1),         # single frame -> jnp.squeeze quirk
            'e': (4, 4, 8, None, 4, 128, 4),     # reference default 4x128
            'f': (3, 3, 8, None, 4, 256, 2),     # BASELINE cfg-2 MLP 4x256
    }.items():
        geo5 = geometry(rng, H, W, G, S)
        ws = random_weights(rng, depth, width, 21)
        t_frames = np.sort(rng.uniform(0.0, 1.0, nt))
        t_inj = -(1000.0 - 6.0)
        pred = network.NeRF_Predictor(8.0, 2.5, 8.0, 4.0, net_depth=depth, net_width=width)

        def predictor_fn(variables, *args, _ws=ws, _pred=pred):
            _dense_params[:] = list(_ws)
            out = _pred(*args)
            assert not _dense_params
            return out

        J = geo5['J'] if S else 1.0
        args = (t_frames, geo5['coords'], geo5['Omega'], J, geo5['g'], geo5['dtau'], geo5['Sigma'],
                0.0, geo5['t_geos'], t_inj)
        emis = predictor_fn({'params': None}, t_frames, units.hr, geo5['coords'], geo5['Omega'], 0.0,
                            geo5['t_geos'], t_inj)
        images = network.image_plane_prediction(None, predictor_fn, *args, units.hr)
        out = dict(emission=emis, images=images)
        for dt in ('full', 'lc'):
            tshape = images.shape if dt == 'full' else images.shape[:-2]
            target = rng.uniform(0, 1e-3, tshape); sigma = rng.uniform(0.5, 2.0, tshape)
            offset = rng.uniform(0, 1e-4, tshape)
            loss, [im2] = network.loss_fn_image(None, predictor_fn, target, sigma, offset, *args, 0.7,
                                                units.hr, dt)
            assert np.array_equal(im2, images)
            out.update({'target_' + dt: target, 'sigma_' + dt: sigma, 'offset_' + dt: offset,
                        'loss_' + dt: loss})
        # G9: finite-difference gradient of the 'full' loss wrt a few weights (float64)
        fd_idx, fd_val = [], []
        for li in range(len(ws)):
            for _ in range(4):
                i = rng.integers(ws[li][0].shape[0]); j = rng.integers(ws[li][0].shape[1])
                h = 1e-6
                vals = []
                for sgn in (+1, -1):
                    ws2 = [(k.copy(), b.copy()) for k, b in ws]
                    ws2[li][0][i, j] += sgn * h
                    l2, _ = network.loss_fn_image(
                        None, lambda v, *a, _w=ws2: predictor_fn(v, *a, _ws=_w), out['target_full'],
                        out['sigma_full'], out['offset_full'], *args, 0.7, units.hr, 'full')
                    vals.append(l2)
                fd_idx.append((li, i, j)); fd_val.append((vals[0] - vals[1]) / (2 * h))
            jb = rng.integers(ws[li][1].shape[0]); vals = []
            for sgn in (+1, -1):
                ws2 = [(k.copy(), b.copy()) for k, b in ws]
                ws2[li][1][jb] += sgn * 1e-6
                l2, _ = network.loss_fn_image(
                    None, lambda v, *a, _w=ws2: predictor_fn(v, *a, _ws=_w), out['target_full'],
                    out['sigma_full'], out['offset_full'], *args, 0.7, units.hr, 'full')
                vals.append(l2)
            fd_idx.append((li, -1, jb)); fd_val.append((vals[0] - vals[1]) / 2e-6)
        out.update(fd_idx=np.array(fd_idx), fd_val=np.array(fd_val))
        for li, (k, b) in enumerate(ws):
            out['kernel%d' % li] = k; out['bias%d' % li] = b
        save('g5_predict_' + tag, t_frames=t_frames, t_injection=t_inj, t_start_obs=0.0,
             hparams=np.array([8.0, 2.5, 8.0, 4.0, 3, depth, width, 0.7]), J=(geo5['J'] if S else np.array(1.0)),
             **{k: geo5[k] for k in ('coords', 'Omega', 't_geos', 'g', 'dtau', 'Sigma')}, **out)

    # G8: image_plane_dynamics / interpolate_coords (voxel forward renderer, emission.py:213-303) -----
    class _Scalar(float):
        data = property(lambda self: float(self))
        def __sub__(self, other):
            return _Scalar(float(self) - float(other))

    class _Coord:
        def __init__(self, v):
            self.v = np.asarray(v)
            self.size = self.v.size
        def max(self):
            return _Scalar(self.v.max())
        def min(self):
            return _Scalar(self.v.min())

    class _FakeDataArray(np.ndarray):          # the three things interpolate_coords asks of an xarray.DataArray
        def __new__(cls, arr, coords):
            obj = np.asarray(arr).view(cls)
            obj.dims = tuple(coords)
            obj._coords = {k: _Coord(v) for k, v in coords.items()}
            return obj
        def __getitem__(self, key):
            if isinstance(key, str):
                return self._coords[key]
            return np.asarray(self).__getitem__(key)

    n = 9
    ax = np.linspace(-6.0, 6.0, n)
    gx, gy, gz = np.meshgrid(ax, ax, ax, indexing='ij')
    vol = np.exp(-((gx - 3.0) ** 2 + (gy + 1.0) ** 2 + gz ** 2) / (2 * 1.2 ** 2)) + 0.1 * rng.uniform(size=gx.shape)
    em0 = _FakeDataArray(vol, {'x': ax, 'y': ax, 'z': ax})
    geo8 = geometry(rng, 4, 5, 12, S=3, rmax=7.0)
    geos = types.SimpleNamespace(x=geo8['coords'][0], y=geo8['coords'][1], z=geo8['coords'][2], t=geo8['t_geos'],
                                 dtau=geo8['dtau'], Sigma=geo8['Sigma'])
    t_frames8 = np.array([0.0, 0.2, 0.55]) * units.hr
    t_inj8 = -(1000.0 - 5.0)
    dyn = emission.image_plane_dynamics(em0, geos, geo8['Omega'], t_frames8, t_inj8, J=1.0, doppler=False)
    dyn_J = emission.image_plane_dynamics(em0, geos, geo8['Omega'], t_frames8, t_inj8, J=geo8['J'], doppler=False)
    dyn_fast = emission.image_plane_dynamics(em0, geos, geo8['Omega'], t_frames8, t_inj8, J=1.0, slow_light=False,
                                             doppler=False, t_start_obs=0.1 * units.hr)
    pts = rng.uniform(-7.0, 7.0, (11, 3))
    interp = emission.interpolate_coords(em0, pts)
    save('g8_dynamics', volume=vol, axis=ax, t_frames=np.asarray(t_frames8.value), t_injection=t_inj8, J=geo8['J'],
         images=np.nan_to_num(dyn, nan=0.0), images_J=np.nan_to_num(dyn_J, nan=0.0), images_fast=np.nan_to_num(dyn_fast, nan=0.0),
         nan_count=np.isnan(dyn).sum(), points=pts, interp=interp,
         **{k: geo8[k] for k in ('coords', 'Omega', 't_geos', 'dtau', 'Sigma')})

    # G7: loss_fn_eht (random complex A) --------------------------------------------------
    nt, H, W, nvis = 3, 4, 4, 7
    images = rng.uniform(0, 1, (nt, H, W))
    fake_pred = lambda *a, **k: None
    orig_ipp = network.image_plane_prediction
    network.image_plane_prediction = lambda *a, **k: images
    try:
        A = rng.normal(size=(nt, nvis, H * W)) + 1j * rng.normal(size=(nt, nvis, H * W))
        A3 = rng.normal(size=(nt, 3, nvis, H * W)) + 1j * rng.normal(size=(nt, 3, nvis, H * W))
        tv = rng.normal(size=(nt, nvis)) + 1j * rng.normal(size=(nt, nvis)); sv = rng.uniform(0.5, 2, (nt, nvis))
        ta = rng.uniform(0, 3, (nt, nvis)); tc = rng.uniform(-np.pi, np.pi, (nt, nvis))
        dummy = [None] * 10
        lv, _ = network.loss_fn_eht(None, fake_pred, tv, sv, A, *dummy, 1.3, None, 'vis')
        la, _ = network.loss_fn_eht(None, fake_pred, ta, sv, A, *dummy, 1.3, None, 'amp')
        lc, _ = network.loss_fn_eht(None, fake_pred, tc, sv, A3, *dummy, 1.3, None, 'cphase')
    finally:
        network.image_plane_prediction = orig_ipp
    save('g7_eht', images=images, A=A, A3=A3, target_vis=tv, target_amp=ta, target_cphase=tc, sigma=sv,
         scale=1.3, loss_vis=lv, loss_amp=la, loss_cphase=lc)

    # G9: GRID_Predictor (network.py:254-353): learnable voxel grid, trilinear sampling (map_coordinates order 1, cval 0),
    # sigmoid(. - 10), domain fill; own rng so that the fixtures above are unaffected ------------------------------------
    rng9 = np.random.default_rng(909)
    res = 7
    geo9 = geometry(rng9, 5, 4, 10, S=None, rmax=7.0)
    grid = rng9.uniform(6.0, 12.0, (res, res, res))                       # sigmoid(grid - 10) spans (0.02, 0.9)
    t_frames9 = np.array([0.0, 0.25, 0.6])
    t_inj9 = -(1000.0 - 7.0)                                             # part of the samples precede the injection
    hp9 = dict(scale=6.0, rmin=1.5, rmax=6.5, z_width=3.0, grid_res=res)
    pred9 = network.GRID_Predictor(**hp9)
    _grid_param[:] = [grid]
    em9 = pred9(t_frames9, units.hr, geo9['coords'], geo9['Omega'], 0.0, geo9['t_geos'], t_inj9)
    img9 = kgeo.radiative_trasfer(em9, geo9['g'], geo9['dtau'], geo9['Sigma'])
    target9 = rng9.uniform(0, 1.0, img9.shape); sigma9 = rng9.uniform(0.5, 2.0, img9.shape)
    def chi2(gr):
        _grid_param[:] = [gr]
        e = pred9(t_frames9, units.hr, geo9['coords'], geo9['Omega'], 0.0, geo9['t_geos'], t_inj9)
        im = kgeo.radiative_trasfer(e, geo9['g'], geo9['dtau'], geo9['Sigma'])
        return float(np.sum(np.abs((im - target9) / sigma9) ** 2))
    fd_idx, fd_val = [], []
    flat_order = rng9.permutation(res ** 3)[:40]
    for f in flat_order:
        i = np.unravel_index(f, grid.shape)
        h = 1e-5
        gp, gm = grid.copy(), grid.copy()
        gp[i] += h; gm[i] -= h
        fd_idx.append(i); fd_val.append((chi2(gp) - chi2(gm)) / (2 * h))
    save('g9_grid', grid=grid, hparams=np.array([hp9['scale'], hp9['rmin'], hp9['rmax'], hp9['z_width'], res]),
         t_frames=t_frames9, t_injection=t_inj9, emission=em9, images=img9, target=target9, sigma=sigma9, loss=chi2(grid),
         fd_idx=np.array(fd_idx), fd_val=np.array(fd_val), nonzero_fd=int((np.abs(np.array(fd_val)) > 0).sum()),
         **{k: geo9[k] for k in ('coords', 'Omega', 't_geos', 'g', 'dtau', 'Sigma')})

    # G10: emission.rotate_evpa (emission.py:395-407) and alma.preprocess_data (alma.py:9-25) on a synthetic two-scan
    # light-curve table (the csv is rebuilt from the stored columns by the test) ------------------------------------------
    import pandas as pd, tempfile
    alma = _load('alma')
    rng10 = np.random.default_rng(1010)
    s2, s3, s4 = rng10.standard_normal((2, 5)), rng10.standard_normal((6, 3, 4)), rng10.standard_normal((3, 2, 4))
    tt = np.concatenate([9.0 + np.arange(500) * 4 / 3600.0, 9.0 + 0.62 + np.arange(450) * 4 / 3600.0])
    lc = pd.DataFrame({'time': tt, 'I': rng10.uniform(2, 3, tt.size), 'Q': rng10.standard_normal(tt.size), 'U': rng10.standard_normal(tt.size)})
    with tempfile.TemporaryDirectory() as d:
        lc.to_csv(os.path.join(d, 'lc.csv'))
        target10, t10 = alma.preprocess_data(os.path.join(d, 'lc.csv'), 12, 0.25, 0.08, 31.0, -17.5, t_start=9.07, t_end=10.0)
    vol = rng10.uniform(0, 2, (4, 5, 6)); vol_est = vol + 0.1 * rng10.standard_normal(vol.shape)
    wc = rng10.uniform(-4, 4, (7, 3))
    extra = dict(vol=vol, vol_est=vol_est, nchw=utils.intensity_to_nchw(vol), nchw_g1=utils.intensity_to_nchw(vol, 'magma', 1.0),
                 mse=utils.mse(vol, vol_est), psnr=utils.psnr(vol, vol_est), wc=wc,
                 wc_img=utils.world_to_image_coords(wc, (8.0, 8.0, 10.0), (16, 16, 20)))
    mv = rng10.uniform(0.1, 1.0, (5, 4, 3, 3)) * np.array([1.0, 0.3, -0.2, 0.05])[None, :, None, None]
    extra.update(stokes_movie=mv.copy(), stokes_norm3=emission.normalize_stokes(mv[:, :3].copy(), 2.5, 0.4),
                 stokes_norm4=emission.normalize_stokes(mv.copy(), 2.5, 0.4, V_flux=-0.1))
    save('g10_alma', **extra, s2=s2, s3=s3, s4=s4, rot2=emission.rotate_evpa(s2, 0.37), rot3=emission.rotate_evpa(s3, -1.2, axis=1),
         rot4=emission.rotate_evpa(s4, 2.9, axis=2), lc_time=tt, lc_I=lc['I'].values, lc_Q=lc['Q'].values, lc_U=lc['U'].values,
         pre_args=np.array([12, 0.25, 0.08, 31.0, -17.5, 9.07, 10.0]), pre_target=target10, pre_t_hr=np.asarray(t10.to('hr').value))


if __name__ == '__main__':
    main()
