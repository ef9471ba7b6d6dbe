"""bhnerf.emission hot-path members: velocity_warp_coords (emission.py:143-211) and
fill_unsupervised_emission (emission.py:343-374).

Inside training/rendering both are fused into the predictor kernel (csrc/fused_common.h,
point_prologue / epilogue); the stand-alone functions below keep the reference API for callers
that use them directly: NumPy in -> NumPy out (the reference's host path), torch tensors in ->
evaluated with torch ops on the tensor's device.
"""
import numpy as np
import torch

from . import constants, units, utils


def velocity_warp_coords(coords, Omega, t_frames, t_start_obs, t_geos, t_injection, rot_axis=[0, 0, 1],
                         M=None, t_units=None, use_jax=False):
    """Warp coordinates by the Keplerian rotation accumulated since injection; NaN before it."""
    xp = utils._xp(coords, Omega, t_geos)
    arr = (lambda v: v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))) if xp is torch \
        else np.asarray
    coords, Omega = arr(coords), arr(Omega)
    if units.is_quantity(t_start_obs):
        t_units = t_start_obs.unit
        t_start_obs = float(t_start_obs.value)
    mass = constants.sgra_mass_msun if M is None else float(getattr(M, 'value', M))
    GM_c3 = constants.GM_c3(t_units, mass) if t_units is not None else 1.0       # emission.py:183-185
    if units.is_quantity(t_frames):
        t_frames = t_frames.to(t_units).value
    t_frames = arr(np.asarray(t_frames, dtype=np.float64)) if not isinstance(t_frames, torch.Tensor) else t_frames
    if xp is torch:
        t_frames = t_frames.to(coords.device, coords.dtype)
        Omega = Omega.to(coords.device)
    if Omega.ndim == 0:                                                           # emission.py:192-193
        Omega = utils.expand_dims(Omega, coords.ndim - 1, axis=-1)
    if t_frames.ndim != 0:                                                        # emission.py:196-198
        coords = utils.expand_dims(coords, coords.ndim + t_frames.ndim, 1)
        t_frames = utils.expand_dims(t_frames, t_frames.ndim + Omega.ndim, -1)
    t_geos = arr(t_geos) if not np.isscalar(t_geos) else t_geos
    if xp is torch and isinstance(t_geos, torch.Tensor):
        t_geos = t_geos.to(coords.device)
    t_M = (t_frames - t_start_obs) / GM_c3 + t_geos - t_injection                 # emission.py:200-201
    theta = t_M * Omega
    nan = float('nan')
    theta = xp.where(t_M < 0.0, xp.full_like(theta, nan), theta)                 # emission.py:204-205
    inv_rot = utils.rotation_matrix(rot_axis, -theta)                             # emission.py:207
    warped = (inv_rot * coords).sum(1)                                            # emission.py:209
    return xp.moveaxis(warped, 0, -1)


def fill_unsupervised_emission(emission, coords, rmin=0, rmax=np.inf, z_width=2.0, fill_value=0.0, use_jax=False):
    """Zero emission outside rmin <= r <= rmax, |z| <= z_width (evaluated on the un-warped coords)."""
    xp = utils._xp(emission, coords if not isinstance(coords, (list, tuple)) else coords[0])
    if xp is torch:
        sq = [torch.squeeze(torch.as_tensor(c)) ** 2 for c in coords]
        r_sq = sq[0] + sq[1] + sq[2]
        z = torch.as_tensor(coords[2])
        fill = torch.full_like(emission, fill_value)
    else:
        r_sq = np.sum(np.array([np.squeeze(c) ** 2 for c in coords]), axis=0)
        z = np.asarray(coords[2])
        fill = np.full_like(emission, fill_value)
    emission = xp.where(r_sq < rmin ** 2, fill, emission)
    emission = xp.where(r_sq > rmax ** 2, fill, emission)
    emission = xp.where(abs(z) > z_width, fill, emission)
    return emission
