"""bhnerf.kgeo hot-path member: the radiative-transfer ray integral (kgeo.py:595-622).

The reference spells it ``radiative_trasfer``; both spellings are exported.  Device tensors go
through the HIP kernel ``bhn_radiative_transfer_fwd``/``_bwd`` (differentiable w.r.t. emission);
NumPy inputs follow the reference's own NumPy path (``use_jax=False``).

The GR pre-compute part of bhnerf/kgeo.py is restated below on plain NumPy arrays (no xarray; 4-vectors
carry their component index ``mu = (t, r, theta, phi)`` on the LAST axis): ``image_plane_geos`` (own ray tracer,
``geodesics.py``), ``wave_vector``, ``spacetime_metric``, ``azimuthal_velocity_vector``, ``doppler_factor``,
``fluid_frame_tetrad``, ``magnetic_field_fluid_frame``, ``parallel_transport`` and the ZAMO variants
(kgeo.py:65-593).  ``geos`` is anything with attribute access to the fields of the reference's geodesic dataset
(``geodesics.Geodesics``, an ``xarray.Dataset`` of the real ``kgeo``, a ``types.SimpleNamespace`` ...).  Parity:
unpinned against the reference (its functions need xarray and the external ``kgeo``, neither is available); pinned
instead by the invariants they must satisfy -- ``u.u = -1``, ``k.k = 0``, orthonormal tetrads, the special-
relativistic Doppler factor far from the hole, conservation of the Penrose-Walker constant along a geodesic
(``tests/test_geodesics_cpu.py``).
"""
import numpy as np
import torch

from . import _hip, utils


def _plane_shape(emission, *vs):
    """Shape (*spatial, G) of the per-ray factors: the first array-valued factor decides, else the
    trailing (H, W, G) axes of the emission."""
    for v in vs:
        if np.ndim(v) > 0:
            return tuple(v.shape)
    return tuple(emission.shape[-3:]) if emission.ndim >= 3 else tuple(emission.shape)


class _RadiativeTransfer(torch.autograd.Function):
    """img[n, r] = sum_k g^2 e[n, r, k] dtau Sigma through the C ABI (fwd and bwd kernels)."""

    @staticmethod
    def forward(ctx, emission, g, dtau, Sigma):
        full = tuple(g.shape)
        G, R = full[-1], int(np.prod(full[:-1]))
        lead = tuple(emission.shape[:emission.ndim - len(full)])
        N = int(np.prod(lead)) if lead else 1
        out = torch.empty(lead + full[:-1], dtype=torch.float32, device=emission.device)
        _hip.check(_hip.lib().bhn_radiative_transfer_fwd(
            _hip.ptr(emission), _hip.ptr(g), _hip.ptr(dtau), _hip.ptr(Sigma), _hip.ptr(out), N, R, G,
            _hip.stream_ptr(emission.device)))
        ctx.save_for_backward(g, dtau, Sigma)
        ctx.dims = (tuple(emission.shape), N, R, G)
        return out

    @staticmethod
    def backward(ctx, dimg):
        g, dtau, Sigma = ctx.saved_tensors
        eshape, N, R, G = ctx.dims
        de = torch.empty(eshape, dtype=torch.float32, device=dimg.device)
        _hip.check(_hip.lib().bhn_radiative_transfer_bwd(
            _hip.ptr(dimg.contiguous()), _hip.ptr(g), _hip.ptr(dtau), _hip.ptr(Sigma), _hip.ptr(de), N, R, G,
            _hip.stream_ptr(dimg.device)))
        return de, None, None, None


def radiative_trasfer(emission, g, dtau, Sigma, use_jax=False):
    """stokes = sum_k g^2 * emission * dtau * Sigma over the last (geodesic-sample) axis."""
    if isinstance(emission, torch.Tensor):
        _hip.require_device(emission)
        full = _plane_shape(emission, g, dtau, Sigma)
        dev = emission.device
        planes = [_hip.as_f32(v, dev).expand(full).contiguous() for v in (g, dtau, Sigma)]
        return _RadiativeTransfer.apply(emission.to(torch.float32).contiguous(), *planes)
    nd = np.ndim(emission)
    g, dtau, Sigma = (utils.expand_dims(v, nd) for v in (g, dtau, Sigma))
    return (g ** 2 * emission * dtau * Sigma).sum(axis=-1)


radiative_transfer = radiative_trasfer


# ---------------------------------------------------------------------------------------------------------------
# GR pre-compute on NumPy arrays
# ---------------------------------------------------------------------------------------------------------------
from .geodesics import Geodesics, image_plane_geos, kerr_functions  # noqa: E402,F401


def _f(geos, name):
    return np.asarray(getattr(geos, name), dtype=np.float64)


def transform_coordinates(v, tetrad, contraction):
    """``tetrad @ v`` over the last axis (kgeo.py:65-90); ``contraction='upper'`` uses the transposed tetrad."""
    v = np.asarray(v)
    if contraction == 'upper':
        tetrad = np.swapaxes(tetrad, -1, -2)
    elif contraction != 'lower':
        raise AttributeError('contraction can be either "upper" or "lower"')
    return np.einsum('...ij,...j->...i', tetrad, v)


def wave_vector(geos):
    """Covariant photon momentum ``k_mu = (-E, +-E sqrt(R)/Delta, +-E sqrt(Theta), E lam)`` (kgeo.py:92-116); the
    signs follow the direction of motion along the affine parameter (turning points)."""
    r, th, aff = _f(geos, 'r'), _f(geos, 'theta'), _f(geos, 'affine')
    E = _f(geos, 'E')
    lam = _f(geos, 'lam')
    if lam.ndim == r.ndim - 1:
        lam = lam[..., None]
    with np.errstate(divide='ignore', invalid='ignore'):
        pm_r = np.sign(np.gradient(r, axis=-1) / np.gradient(aff, axis=-1))
        pm_th = np.sign(np.gradient(th, axis=-1) / np.gradient(aff, axis=-1))
    k_t = -E * np.ones_like(r)
    k_r = E * np.sqrt(np.clip(_f(geos, 'R'), 0.0, None)) * pm_r / _f(geos, 'Delta')
    k_th = E * np.sqrt(np.clip(_f(geos, 'Theta'), 0.0, None)) * pm_th
    k_ph = E * lam * np.ones_like(r)
    return np.stack([k_t, k_r, k_th, k_ph], axis=-1)


def spacetime_metric(geos):
    """Non-zero covariant Kerr metric components in Boyer-Lindquist coordinates (kgeo.py:118-144)."""
    r, th, M, a = _f(geos, 'r'), _f(geos, 'theta'), _f(geos, 'M'), _f(geos, 'spin')
    Sg, Dl, Xi = _f(geos, 'Sigma'), _f(geos, 'Delta'), _f(geos, 'Xi')
    s2 = np.sin(th) ** 2
    return Geodesics(tt=-(1.0 - 2.0 * M * r / Sg), rr=Sg / Dl, thth=Sg, phph=Xi * s2 / Sg, tph=-2.0 * M * a * r * s2 / Sg)


def spacetime_inv_metric(geos):
    """Non-zero contravariant components (kgeo.py:146-173)."""
    r, th, M, a = _f(geos, 'r'), _f(geos, 'theta'), _f(geos, 'M'), _f(geos, 'spin')
    Sg, Dl, Xi = _f(geos, 'Sigma'), _f(geos, 'Delta'), _f(geos, 'Xi')
    s2 = np.sin(th) ** 2
    return Geodesics(tt=-Xi / (Dl * Sg), rr=Dl / Sg, thth=1.0 / Sg, phph=(Dl - a ** 2 * s2) / (Dl * Sg * s2),
                     tph=-2.0 * M * a * r / (Dl * Sg))


def raise_or_lower_indices(g, u):
    """``g_{mu nu} u^nu`` (or the inverse) for the block-diagonal + t-phi metric (kgeo.py:175-197)."""
    u = np.asarray(u)
    return np.stack([g.tt * u[..., 0] + g.tph * u[..., 3], g.rr * u[..., 1], g.thth * u[..., 2],
                     g.phph * u[..., 3] + g.tph * u[..., 0]], axis=-1)


def azimuthal_velocity_vector(geos, Omega):
    """Contravariant 4-velocity of matter on circular orbits with angular velocity ``Omega`` (kgeo.py:199-223)."""
    g = spacetime_metric(geos)
    Omega = np.asarray(Omega, dtype=np.float64) * np.ones_like(g.tt)
    with np.errstate(invalid='ignore', divide='ignore'):              # superluminal Omega -> NaN, as in the reference
        ut = 1.0 / np.sqrt(-(g.tt + 2.0 * Omega * g.tph + g.phph * Omega ** 2))
    zero = np.zeros_like(ut)
    return np.stack([ut, zero, zero, ut * Omega], axis=-1)


def doppler_factor(geos, umu, fillna=0.0):
    """``g = E / -(k_mu u^mu)`` (kgeo.py:225-248); NaNs (superluminal Omega) are replaced by ``fillna`` unless it is
    False or None."""
    g = _f(geos, 'E') / -(wave_vector(geos) * np.asarray(umu)).sum(axis=-1)
    if not ((isinstance(fillna, bool) and fillna is False) or fillna is None):
        g = np.where(np.isnan(g), fillna, g)
    return g


def magnetic_field_spherical(geos, b_r, b_th, b_ph):
    """A (r, theta, phi) field sampled on the geodesics; scalars are broadcast (kgeo.py:250-272)."""
    one = np.ones_like(_f(geos, 'r'))
    return np.stack([np.asarray(b_r) * one, np.asarray(b_th) * one, np.asarray(b_ph) * one], axis=-1)


def fluid_frame_tetrad(geos, umu):
    """Tetrad ``e[..., mu, a]`` (component index, then leg: the reference's layout) of the frame co-moving with
    ``umu`` (kgeo.py:310-345)."""
    g = spacetime_metric(geos)
    umu = np.asarray(umu)
    u_mu = raise_or_lower_indices(g, umu)
    uu = u_mu * umu
    th, Dl = _f(geos, 'theta'), _f(geos, 'Delta')
    tph = uu[..., 0] + uu[..., 3]
    N_r = np.sqrt(-g.rr * tph * (1.0 + uu[..., 2]))
    N_th = np.sqrt(g.thth * (1.0 + uu[..., 2]))
    N_ph = np.sqrt(-tph * Dl * np.sin(th) ** 2)
    zero = np.zeros_like(N_r)
    e_t = -umu
    e_r = np.stack([u_mu[..., 1] * umu[..., 0], -tph, zero, u_mu[..., 1] * umu[..., 3]], axis=-1) / N_r[..., None]
    e_th = np.stack([u_mu[..., 2] * umu[..., 0], u_mu[..., 2] * umu[..., 1], 1.0 + uu[..., 2], u_mu[..., 2] * umu[..., 3]],
                    axis=-1) / N_th[..., None]
    e_ph = np.stack([u_mu[..., 3], zero, zero, -u_mu[..., 0]], axis=-1) / N_ph[..., None]
    return np.stack([e_t, e_r, e_th, e_ph], axis=-1)


def magnetic_field_fluid_frame(geos, umu, arad, avert, ator):
    """Lab-frame field (radial / vertical / toroidal amplitudes) seen in the fluid frame (kgeo.py:274-308)."""
    th = _f(geos, 'theta')
    Br = arad * np.sin(th) + avert * np.cos(th)
    Bth = avert * (-np.sin(th))
    Bph = ator * np.ones_like(th)
    g = spacetime_metric(geos)
    umu = np.asarray(umu)
    u_mu = raise_or_lower_indices(g, umu)
    e = fluid_frame_tetrad(geos, umu)
    b0 = Br * u_mu[..., 1] + Bth * u_mu[..., 2] + Bph * u_mu[..., 3]
    b1 = (Br + b0 * u_mu[..., 1]) / u_mu[..., 0]
    b2 = (Bth + b0 * u_mu[..., 2]) / u_mu[..., 0]
    b3 = (Bph + b0 * u_mu[..., 3]) / u_mu[..., 0]
    b_mu = raise_or_lower_indices(g, np.stack([b0, b1, b2, b3], axis=-1))
    return transform_coordinates(b_mu, e, 'upper')[..., 1:]


def zamo_frame_velocity(geos, beta, chi):
    """Velocity of an observer boosted by ``beta`` at angle ``chi`` relative to the ZAMO (kgeo.py:408-436)."""
    r, Dl, Xi, om = _f(geos, 'r'), _f(geos, 'Delta'), _f(geos, 'Xi'), _f(geos, 'omega')
    gam = 1.0 / np.sqrt(1.0 - beta ** 2)
    ut = (gam / r) * np.sqrt(Xi / Dl)
    ur = (beta * gam * np.cos(chi) / r) * np.sqrt(Dl)
    uph = ut * om + r * beta * gam * np.sin(chi) / np.sqrt(Xi)
    return np.stack([ut, ur, np.zeros_like(ut), uph], axis=-1)


def zamo_frame_tetrad(geos, beta, chi):
    """Tetrad of the boosted ZAMO frame, Gelles et al. 2021 eq. A4 (kgeo.py:347-406)."""
    r, Dl, Xi, om = _f(geos, 'r'), _f(geos, 'Delta'), _f(geos, 'Xi'), _f(geos, 'omega')
    gam = 1.0 / np.sqrt(1.0 - beta ** 2)
    c, s = np.cos(chi), np.sin(chi)
    sxd, sd, sx = np.sqrt(Xi / Dl), np.sqrt(Dl), np.sqrt(Xi)
    zero = np.zeros_like(r)
    e_t = np.stack([(gam / r) * sxd, (beta * gam * c / r) * sd, zero, (gam * om / r) * sxd + r * beta * gam * s / sx], axis=-1)
    e_r = np.stack([(beta * gam * c / r) * sxd, ((1.0 + (gam - 1.0) * c ** 2) / r) * sd, zero,
                    beta * gam * om * c / r * sxd + r * (gam - 1.0) * c * s / sx], axis=-1)
    e_th = np.stack([zero, zero, 1.0 / r, zero], axis=-1)
    e_ph = np.stack([(beta * gam * s / r) * sxd, ((gam - 1.0) * c * s / r) * sd, zero,
                     beta * om * s * (gam / r) * sxd + r * ((gam - 1.0) * s ** 2 + 1.0) / sx], axis=-1)
    return np.stack([e_t, e_r, e_th, e_ph], axis=-1)


def _transport(geos, e_mu, g, b, Q_frac, V_frac, spectral_index, with_v):
    """Common part of parallel_transport / parallel_transport_zamo (kgeo.py:438-593): local EVPA from k x B in the
    emitter frame, emissivity scalings, rotation to the observer screen through the Penrose-Walker constant
    (Himwich et al. 2020)."""
    if Q_frac > 1.0 or Q_frac < 0.0:
        raise AttributeError('Q_frac should be in [0,1]')
    b = np.asarray(b, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    k_mu = wave_vector(geos)
    k_loc = transform_coordinates(k_mu, e_mu, 'upper')[..., 1:]
    k_mag = np.sqrt((k_loc ** 2).sum(axis=-1))
    f_loc = np.cross(k_loc, b, axis=-1) / k_mag[..., None]
    f_glob = transform_coordinates(np.concatenate([np.zeros_like(f_loc[..., :1]), f_loc], axis=-1), e_mu, 'lower')
    ft, fr, fth, fph = (f_glob[..., i] for i in range(4))
    b_mag = np.sqrt((b ** 2).sum(axis=-1))
    sin_b = np.sqrt((f_loc ** 2).sum(axis=-1)) / k_mag
    I = g ** spectral_index * b_mag ** (spectral_index + 1) * sin_b ** (spectral_index + 1)
    Q = Q_frac * I
    r, th, a = _f(geos, 'r'), _f(geos, 'theta'), _f(geos, 'spin')
    kup = raise_or_lower_indices(spacetime_inv_metric(geos), k_mu)
    s2 = np.sin(th) ** 2
    A = (kup[..., 0] * fr - kup[..., 1] * ft) + a * s2 * (kup[..., 1] * fph - kup[..., 3] * fr)
    B = ((r ** 2 + a ** 2) * (kup[..., 3] * fth - kup[..., 2] * fph) - a * (kup[..., 0] * fth - kup[..., 2] * ft)) * np.sin(th)
    kappa = (r - 1j * a * np.cos(th)) * (A - 1j * B)
    alpha, beta = _f(geos, 'alpha'), _f(geos, 'beta')
    if alpha.ndim == r.ndim - 1:
        alpha, beta = alpha[..., None], beta[..., None]
    mu = -(alpha + a * np.sin(_f(geos, 'inc')))
    with np.errstate(divide='ignore', invalid='ignore'):
        chi2 = np.angle(((beta + 1j * mu) * kappa.conj()) / ((beta - 1j * mu) * kappa))
    J = [I, np.cos(chi2) * Q, np.sin(chi2) * Q]                          # rot(chi2) applied to (Q, U = 0)
    if with_v:
        with np.errstate(divide='ignore', invalid='ignore'):
            cot_b = np.sqrt(1.0 - sin_b ** 2) / sin_b
            J.append(V_frac * g ** (-spectral_index - 0.5) * b_mag ** (spectral_index + 1.5) * sin_b ** (spectral_index + 1.5) * cot_b)
    return np.stack(J), kappa


def parallel_transport(geos, umu, g, b, Q_frac=0.2, V_frac=0.01, spectral_index=1):
    """Stokes scaling factors ``J = [I, Q, U(, V)]`` transported to the observer screen (kgeo.py:438-519); the V row
    is dropped when ``V_frac == 0``."""
    J, _ = _transport(geos, fluid_frame_tetrad(geos, umu), g, b, Q_frac, V_frac, spectral_index, V_frac != 0)
    return J


def parallel_transport_zamo(geos, beta_v, chi, g, b, Q_frac=0.2, spectral_index=1):
    """The same in the boosted ZAMO frame, ``J = [I, Q, U]`` (kgeo.py:521-593)."""
    J, _ = _transport(geos, zamo_frame_tetrad(geos, beta_v, chi), g, b, Q_frac, 0.0, spectral_index, False)
    return J
