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


class Volume(np.ndarray):
    """Gridded field with named axes -- the parts of ``xarray.DataArray`` this package and the reference's drivers touch
    (xarray is not a dependency): ``dims``, coordinate arrays by name (``vol['x']``, ``vol.x``), ``data``, ``attrs``,
    ``integrate``.  Accepted wherever the reference takes an emission DataArray (``emission.image_plane_dynamics``,
    ``interpolate_coords``, ``SummaryWriter.recovery_3d``)."""

    def __new__(cls, data, coords, dims, attrs=None):
        obj = np.asarray(data).view(cls)
        obj.dims, obj.coords, obj.attrs = tuple(dims), {d: np.asarray(coords[d]) for d in dims}, dict(attrs or {})
        return obj

    def __array_finalize__(self, src):
        self.dims = getattr(src, 'dims', ())
        self.coords = getattr(src, 'coords', {})
        self.attrs = dict(getattr(src, 'attrs', {}))

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        out = super().__getitem__(key)
        return out.view(np.ndarray) if isinstance(out, Volume) else out       # a slice no longer matches the coordinates

    def __getattr__(self, name):                     # vol.x, vol.y, vol.z
        coords = self.__dict__.get('coords', {})
        if name in coords:
            return coords[name]
        raise AttributeError(name)

    @property
    def data(self):
        return self.view(np.ndarray)

    def integrate(self, dims):
        """Trapezoidal integral over the named axes (``DataArray.integrate``)."""
        out = self.view(np.ndarray)
        for d in sorted((self.dims.index(d) for d in np.atleast_1d(dims)), reverse=True):
            out = (np.trapezoid if hasattr(np, 'trapezoid') else np.trapz)(out, self.coords[self.dims[d]], axis=d)
        return out


def linspace_xr(num, start=-0.5, stop=0.5, endpoint=True, units='unitless'):
    """Coordinates ``x[, y[, z]]`` linearly spaced over ``[start, stop]`` (utils.py:15-46), as a ``{dim: array}`` dict."""
    return {d: np.linspace(start, stop, int(n), endpoint=endpoint) for d, n in zip(('x', 'y', 'z'), np.atleast_1d(num))}


def gaussian_xr(resolution, center, std, fov=(1.0, 'unitless'), std_clip=np.inf):
    """Gaussian blob on a regular grid of extent ``fov[0]`` centred on the origin (utils.py:48-95); values below
    ``exp(-std_clip^2 / 2)`` are zeroed.  3-D: dims (x, y, z); 2-D: dims (y, x) as in the reference."""
    resolution = np.atleast_1d(resolution)
    std = (std,) * 3 if np.isscalar(std) else tuple(std)
    if len(resolution) != len(center):
        raise AttributeError('resolution and center should have same length {} != {}'.format(len(resolution), len(center)))
    grid = linspace_xr(resolution, start=-fov[0] / 2.0, stop=fov[0] / 2.0, units=fov[1])
    if len(resolution) == 3:
        dims = ('x', 'y', 'z')
        arg = (((grid['x'] - center[0]) / std[0]) ** 2)[:, None, None] + (((grid['y'] - center[1]) / std[1]) ** 2)[None, :, None] \
            + (((grid['z'] - center[2]) / std[2]) ** 2)[None, None, :]
    elif len(resolution) == 2:
        dims = ('y', 'x')
        arg = (((grid['y'] - center[1]) / std[1]) ** 2)[:, None] + (((grid['x'] - center[0]) / std[0]) ** 2)[None, :]
    else:
        raise AttributeError
    data = np.exp(-0.5 * arg)
    data = np.where(data > np.exp(-0.5 * std_clip ** 2), data, 0.0)
    return Volume(data, grid, dims, dict(fov=fov, std=std, center=center, std_clip=std_clip))
