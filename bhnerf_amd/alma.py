"""ALMA light-curve helpers: the callers either side of the hot path for polarised light-curve fits
(reference: bhnerf/alma.py).

Everything here is host-side preparation (geodesics, Doppler factor, polarised transport factors -> ``raytracing_args``)
or post-processing (chi-square of a checkpoint against the light curves).  The rendering itself goes through
``network.image_plane_checkpoint`` -> HIP forward kernel.
"""
import os

import numpy as np

from . import constants, emission, kgeo, network, units

_ROTATION_SIGN = {'cw': -1.0, 'ccw': 1.0}
_STOKES = ('I', 'Q', 'U')


def preprocess_data(data_path, window_size, I_hs_mean, P_sha, chi_sha, de_rot_angle, t_start=9.33, t_end=11.05):
    """Window-averaged, shadow-subtracted, de-rotated (I, Q, U) light-curve targets and their times (alma.py:9-25).

    ``data_path`` is a csv with columns ``time`` [UTC hr], ``Q``, ``U``.  Rows of the ``[t_start, t_end]`` period are
    averaged in consecutive windows of ``window_size`` samples; windows that straddle two scans (time jump of 160 s or
    more) are dropped.  The constant accretion-disk polarisation ``P_sha exp(2i chi_sha)`` is subtracted, the Faraday
    rotation undone (``de_rot_angle`` [deg]) and a constant hot-spot intensity ``I_hs_mean`` prepended."""
    import pandas as pd
    lc = pd.read_csv(data_path, index_col=0)
    lc = lc.loc[(lc['time'] >= t_start) & (lc['time'] <= t_end)]
    means = lc.rolling(window_size).mean().loc[::window_size].dropna()
    means = means.where(means['time'].diff().fillna(0.0) < 160.0 / 3600.0).dropna()
    t_frames = means['time'].values * units.hr

    chi = np.deg2rad(chi_sha)
    shadow_qu = P_sha * np.array([np.cos(2.0 * chi), np.sin(2.0 * chi)])
    qu = emission.rotate_evpa(np.asarray(means[['Q', 'U']], dtype=np.float64) - shadow_qu, np.deg2rad(de_rot_angle), axis=1)
    target = np.concatenate([np.full((qu.shape[0], 1), float(I_hs_mean)), qu], axis=1)
    return target, t_frames


def image_plane_model(inc, spin, params, rot_angle=0.0, randomize_subpixel_rays=False):
    """Geodesics, Keplerian angular velocity and polarised emission factors J (3, alpha, beta, geo) of one
    (inclination, spin) hypothesis (alma.py:27-64).

    ``params``: ``num_alpha, num_beta, fov_M, z_width, rmin`` (a radius or ``'ISCO'``), ``Q_frac``, ``b_consts``
    (``arad, avert, ator``), ``Omega_dir`` (``'cw'|'ccw'``) and optionally ``Omega_frac`` (sub-Keplerian factor)."""
    fov = params['fov_M']
    rmin = float(constants.isco_pro(spin)) if params['rmin'] == 'ISCO' else params['rmin']
    rmax = fov / 2.0
    geos = kgeo.image_plane_geos(spin, inc, num_alpha=params['num_alpha'], num_beta=params['num_beta'],
                                 alpha_range=[-fov / 2.0, fov / 2.0], beta_range=[-fov / 2.0, fov / 2.0],
                                 randomize_subpixel_rays=randomize_subpixel_rays).fillna(0.0)

    # prograde / retrograde Keplerian rotation, Doppler boosting
    sqrt_M = np.sqrt(geos.M)
    with np.errstate(divide='ignore', invalid='ignore'):
        Omega = params.get('Omega_frac', 1.0) * _ROTATION_SIGN[params['Omega_dir']] * sqrt_M / (geos.r ** 1.5 + geos.spin * sqrt_M)
    umu = kgeo.azimuthal_velocity_vector(geos, Omega)
    g = kgeo.doppler_factor(geos, umu)

    # fluid-frame magnetic field, normalised to unit mean magnitude over the recovery domain
    b = kgeo.magnetic_field_fluid_frame(geos, umu, **params['b_consts'])
    domain = (np.abs(geos.z) < params['z_width']) & (geos.r > rmin) & (geos.r < rmax)
    b = b / np.sqrt((b[domain] ** 2).sum(axis=-1)).mean()

    # polarised emission factors with parallel transport to the observer's screen
    with np.errstate(divide='ignore', invalid='ignore'):
        J = kgeo.parallel_transport(geos, umu, g, b, Q_frac=params['Q_frac'], V_frac=0)
    J = emission.rotate_evpa(np.nan_to_num(J, nan=0.0), rot_angle)
    return geos, Omega, J


def get_raytracing_args(inc, spin, params, stokes=_STOKES, rot_angle=0.0, num_subpixel_rays=1):
    """List of ``network.raytracing_args`` -- one per sub-pixel ray set (alma.py:66-82).  One set uses the regular pixel
    centres; several sets jitter the rays inside their pixels."""
    rows = [_STOKES.index(s) for s in stokes]
    jitter = num_subpixel_rays != 1
    out = []
    for _ in range(num_subpixel_rays):
        geos, Omega, J = image_plane_model(inc, spin, params, rot_angle, jitter)
        t_injection = -float(geos.r_o + params['fov_M'] / 4.0)
        out.append(network.raytracing_args(geos, Omega, t_injection, params['t_start_obs'] * units.hr, J[rows]))
    return out


def chi2_lightcurves(raytracing_args, checkpoint_dir, t, data, sigma=1.0, rmin=0.0, rmax=np.inf, batchsize=20, **predictor_kw):
    """Chi-square per frame of the light curves rendered from the newest checkpoint (alma.py:84-87).  ``predictor_kw``
    (``mode``, ``device``) goes to the predictor rebuilt from the checkpoint directory."""
    image_plane = network.image_plane_checkpoint(raytracing_args, checkpoint_dir, t, rmin, rmax, batchsize, **predictor_kw)
    lightcurves = np.asarray(image_plane).sum(axis=(-1, -2))
    return float(np.sum(((lightcurves - data) / sigma) ** 2) / len(t))


def chi2_df(inclinations, spins, seeds, params, checkpoint_fmt, t, data, stokes=_STOKES, sigma=1.0, rot_angle=0.0,
            num_subpixel_rays=1, final_step=50000):
    """Table of chi-square values over an inclination [deg] or a spin sweep x seeds (alma.py:89-117).  Checkpoint
    directories are ``checkpoint_fmt.format(index, seed)``; runs without a ``checkpoint_<final_step>`` stay NaN."""
    import pandas as pd
    inclinations, spins = np.atleast_1d(inclinations), np.atleast_1d(spins)
    if len(inclinations) > 1 and len(spins) > 1:
        raise AttributeError('not implemented')
    if len(spins) > 1:
        index_name, indices = 'spin', spins
        inclinations = np.full_like(spins, inclinations[0], dtype=np.float64)
    else:
        index_name, indices = 'inc', inclinations
        spins = np.full_like(inclinations, spins[0], dtype=np.float64)

    fit = np.full((len(indices), len(seeds)), np.nan)
    traced = (None, None, None)                           # geodesics are shared by the seeds of one (inc, spin)
    for i, (inc, spin) in enumerate(zip(inclinations, spins)):
        for j, seed in enumerate(seeds):
            checkpoint_dir = checkpoint_fmt.format(indices[i], seed)
            if not os.path.exists(os.path.join(checkpoint_dir, 'checkpoint_%d' % final_step)):
                continue
            if traced[:2] != (inc, spin):
                traced = (inc, spin, get_raytracing_args(np.deg2rad(inc), spin, params, stokes, rot_angle, num_subpixel_rays))
            fit[i, j] = chi2_lightcurves(traced[2], checkpoint_dir, t, data, sigma)
    df = pd.DataFrame(fit, index=indices, columns=['seed {}'.format(s) for s in seeds])
    df.index.name = index_name
    return df
