"""The two bhnerf.utils helpers that sit on the hot path (utils.py:97-132, 215-219).

Array-library generic: NumPy arrays in -> NumPy out (the reference's ``use_jax=False`` host
path); torch tensors in -> torch out on the tensor's device.
"""
import numpy as np
import torch


def _xp(*arrays):
    return torch if any(isinstance(a, torch.Tensor) for a in arrays) else np


def expand_dims(x, ndim, axis=0, use_jax=False):
    """Insert unit axes at `axis` until x.ndim == ndim (utils.py:215-219)."""
    if isinstance(x, torch.Tensor):
        while x.ndim < ndim:
            x = x.unsqueeze(min(axis, x.ndim) if axis >= 0 else axis)
        return x
    x = np.asarray(x)
    while x.ndim < ndim:
        x = np.expand_dims(x, axis=min(axis, x.ndim))
    return x


def rotation_matrix(axis, angle, use_jax=False):
    """Euler-Rodrigues rotation about `axis` by `angle` (utils.py:97-132): shape (3,3,*angle.shape)."""
    xp = _xp(angle)
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / np.sqrt(np.dot(ax, ax))
    half = angle / 2.0
    a = xp.cos(half) if xp is torch else np.cos(half)
    sn = xp.sin(half) if xp is torch else np.sin(half)
    b, c, d = -ax[0] * sn, -ax[1] * sn, -ax[2] * sn
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    rows = [[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
            [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
            [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]]
    if xp is torch:
        return torch.stack([torch.stack(r) for r in rows])
    return np.array(rows)


def mse(true, est):
    """Mean squared error (utils.py:9)."""
    return float(np.mean((np.asarray(true) - np.asarray(est)) ** 2))


def psnr(true, est):
    """Peak signal-to-noise ratio in dB relative to ``max(true)`` (utils.py:11)."""
    return float(10.0 * np.log10(np.max(true) ** 2 / mse(true, est)))


def world_to_image_coords(coords, fov, npix, use_jax=False):
    """World coordinates in ``[-fov/2, fov/2]`` -> fractional pixel indices ``[0, npix-1]`` per axis (utils.py:160-166)."""
    xp = _xp(coords)
    return xp.stack([(coords[..., i] + fov[i] / 2.0) / fov[i] * (npix[i] - 1) for i in range(coords.shape[-1])], -1)


def intensity_to_nchw(intensity, cmap='viridis', gamma=0.5):
    """Colour-mapped slices of a 3-D intensity volume for image logging (utils.py:168-193): min-max normalised,
    gamma-corrected, RGB from the matplotlib colour map; the volume axes (0, 1, 2) + colour become axes (3, 2, 0, 1)."""
    import matplotlib
    intensity = np.asarray(intensity, dtype=np.float64)
    lo, hi = intensity.min(), intensity.max()
    with np.errstate(invalid='ignore', divide='ignore'):
        normed = ((intensity - lo) / (hi - lo)) ** gamma
    rgb = matplotlib.colormaps[cmap](normed)[..., :3]
    return np.moveaxis(rgb, (0, 1, 2, 3), (3, 2, 0, 1))
