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


# ------------------------------------------------------------------------------------------------
# Voxel forward renderer
# ------------------------------------------------------------------------------------------------
def _grid_of(emission):
    """(values float32 ndarray, extents [fx,fy,fz]) of an xarray-like 3-D/4-D emission (``.dims`` + coordinate
    arrays, the only things interpolate_coords uses: emission.py:229-230) or of a ``(array, fov)`` tuple."""
    if isinstance(emission, tuple):
        arr, fov = emission
        fov = [float(fov)] * 3 if np.isscalar(fov) else [float(f) for f in fov]
        return np.ascontiguousarray(arr, dtype=np.float32), fov
    dims = list(emission.dims)[-3:]
    fov = [float(np.max(np.asarray(emission[d])) - np.min(np.asarray(emission[d]))) for d in dims]
    return np.ascontiguousarray(np.asarray(getattr(emission, 'data', emission)), dtype=np.float32), fov


def interpolate_coords(emission, coords):
    """Interpolate a 3-D emission field at world coordinates (emission.py:213-233): trilinear, zero
    outside the grid.  coords (..., 3) NumPy -> NumPy via the HIP kernel when a GPU is present, else the
    reference's own host path (scipy.ndimage.map_coordinates); torch device tensors -> device tensor."""
    import ctypes as C
    from . import _hip
    arr, fov = _grid_of(emission)
    if isinstance(coords, torch.Tensor) or torch.cuda.is_available():
        dev = coords.device if isinstance(coords, torch.Tensor) else torch.device('cuda', torch.cuda.current_device())
        pts = _hip.as_f32(coords, dev).reshape(-1, 3).contiguous()
        grid = torch.as_tensor(arr, device=dev)
        out = torch.empty((pts.shape[0],), dtype=torch.float32, device=dev)
        _hip.check(_hip.lib().bhn_trilinear(_hip.ptr(pts), pts.shape[0], _hip.ptr(grid), *arr.shape[-3:], (C.c_float * 3)(*fov),
                                            _hip.ptr(out), _hip.stream_ptr(dev)))
        out = out.reshape(tuple(np.shape(coords)[:-1]))
        return out if isinstance(coords, torch.Tensor) else out.cpu().numpy()
    import scipy.ndimage
    npix = arr.shape
    idx = np.moveaxis(np.stack([(np.asarray(coords)[..., i] + fov[i] / 2.0) / fov[i] * (npix[i] - 1) for i in range(3)], -1), -1, 0)
    return scipy.ndimage.map_coordinates(arr, idx, order=1, cval=0.)


def image_plane_dynamics(emission_0, geos, Omega, t_frames, t_injection, J=1.0, t_start_obs=None, slow_light=True,
                         doppler=True, rot_axis=[0, 0, 1], M=None):
    """Image-plane movie of an initial 3-D emission advected by the velocity field (emission.py:235-303): warp ->
    trilinear sampling -> x J -> radiative transfer, fused in one HIP kernel (``bhn_voxel_render_fwd``).
    ``geos`` carries x, y, z, t, dtau, Sigma; with ``doppler=True`` the Doppler factor is derived from traced geodesics
    (kgeo.azimuthal_velocity_vector / doppler_factor) or taken from ``geos.g``.  Returns a NumPy movie (nt,[S],H,W)."""
    import ctypes as C
    from . import _hip, engine
    if list(rot_axis) != [0, 0, 1]:
        raise AttributeError('only equatorial-plane rotation (rot_axis=[0,0,1]) is supported')
    get = (lambda k: geos[k]) if isinstance(geos, dict) else (lambda k: getattr(geos, k))
    dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise _hip.HipError('image_plane_dynamics runs on the HIP device only (no CPU fallback)')
    coords = np.array([np.asarray(get(k), dtype=np.float32) for k in ('x', 'y', 'z')])
    t_geos = np.asarray(get('t'), dtype=np.float32) if slow_light else 0.0            # emission.py:269
    if doppler:
        def has(k):
            try:
                get(k)
                return True
            except (KeyError, AttributeError):
                return False
        if all(has(k) for k in ('r', 'theta', 'affine', 'R', 'Theta', 'Delta', 'Xi', 'lam', 'E', 'M', 'spin')):
            from . import kgeo                                                       # emission.py:281-282
            g = np.asarray(kgeo.doppler_factor(geos, kgeo.azimuthal_velocity_vector(geos, Omega)), dtype=np.float32)
        elif has('g'):
            g = np.asarray(get('g'), dtype=np.float32)
        else:
            raise AttributeError('doppler=True needs traced geodesics (kgeo.image_plane_geos) or a Doppler factor geos.g')
    else:
        g = 1.0
    if t_start_obs is None:                                                            # emission.py:274 (a scalar frame time is its own start)
        if units.is_quantity(t_frames):
            t_start_obs = float(np.atleast_1d(np.asarray(t_frames.value))[0]) * t_frames.unit
        else:
            t_start_obs = np.atleast_1d(t_frames)[0]
    t_units = t_start_obs.unit if units.is_quantity(t_start_obs) else (t_frames.unit if units.is_quantity(t_frames) else None)
    mass = constants.sgra_mass_msun if M is None else float(getattr(M, 'value', M))
    GM_c3 = constants.GM_c3(t_units, mass) if t_units is not None else 1.0
    tf = np.atleast_1d(np.asarray(units.strip(t_frames, t_units), dtype=np.float64))
    t0 = float(np.asarray(units.strip(t_start_obs, t_units)))
    Jn = None if (np.ndim(J) == 0 and float(J) == 1.0) else J
    geom = engine.RayGeometry(coords, Omega, g, get('dtau'), get('Sigma'), t_geos, Jn, 0.0, np.inf, np.inf, dev)
    arr, fov = _grid_of(emission_0)
    if arr.ndim == 4 and arr.shape[0] != tf.size:
        raise AttributeError('a 4-D emission needs one grid per frame')
    grid = torch.as_tensor(arr, device=dev)
    tM0 = engine.frame_offsets(tf, t0, t_injection, GM_c3, dev)
    images = torch.empty((tf.size, geom.Sx, geom.R), dtype=torch.float32, device=dev)
    gs, fs = geom.c_struct(), _hip.bhn_frames(int(tf.size), tM0.data_ptr())
    nx, ny, nz = arr.shape[-3:]
    _hip.check(_hip.lib().bhn_voxel_render_fwd(C.byref(gs), C.byref(fs), _hip.ptr(grid), nx, ny, nz,
                                               nx * ny * nz if arr.ndim == 4 else 0, (C.c_float * 3)(*fov), _hip.ptr(images),
                                               _hip.stream_ptr(dev)))
    out = images.reshape((tf.size,) + ((geom.S,) if geom.S else ()) + geom.spatial).cpu().numpy()
    if np.ndim(t_frames if not units.is_quantity(t_frames) else t_frames.value) == 0:
        out = out[0]
    return np.squeeze(out) if geom.S else out                                          # emission.py:299 squeeze quirk


def rotate_evpa(stokes, angle, axis=0):
    """Rotate the electric-vector position angle by ``angle`` [rad]: ``Q + iU -> exp(2i angle) (Q + iU)``
    (emission.py:395-407).  The Stokes axis holds (Q, U), (I, Q, U) or (I, Q, U, V); I and V are unchanged."""
    stokes = np.asarray(stokes)
    n = stokes.shape[axis]
    if n not in (2, 3, 4):
        raise AttributeError('Shape of stokes vector along axis={} not supported'.format(axis))
    q = 0 if n == 2 else 1
    moved = np.moveaxis(stokes, axis, 0)
    c, s = np.cos(2.0 * angle), np.sin(2.0 * angle)
    out = np.array(moved, dtype=np.result_type(moved.dtype, np.float64), copy=True)
    out[q] = c * moved[q] - s * moved[q + 1]
    out[q + 1] = s * moved[q] + c * moved[q + 1]
    return np.moveaxis(out, 0, axis)


def generate_hotspot_xr(resolution, rot_axis, rot_angle, orbit_radius, std, r_isco, fov, std_clip=np.inf, normalize=True):
    """Gaussian hotspot at angle ``rot_angle`` of a circular orbit of radius ``orbit_radius`` about ``rot_axis``
    (emission.py:10-60), as a ``utils.Volume``; ``normalize`` divides by the volume integral."""
    if orbit_radius < r_isco:
        raise AttributeError('hotspot center ({}) is is within r_isco: {}'.format(orbit_radius, r_isco))
    resolution = np.atleast_1d(resolution)
    center = orbit_radius * np.array([np.cos(rot_angle), np.sin(rot_angle)])
    if len(resolution) != 2:
        center = np.matmul(_tilt_to_axis(rot_axis), np.append(center, 0.0))
    vol = utils.gaussian_xr(resolution, center, std, fov=fov, std_clip=std_clip)
    if normalize:
        vol = utils.Volume(vol.data / vol.integrate(list(vol.dims) if len(resolution) == 2 else ['x', 'y', 'z']), vol.coords, vol.dims, vol.attrs)
    vol.attrs.update(rot_axis=rot_axis)
    return vol


def _tilt_to_axis(rot_axis):
    """Rotation taking the z axis onto ``rot_axis`` (the orbit-plane tilt of emission.py:46-52, 85-91)."""
    axis = np.asarray(rot_axis, dtype=np.float64)
    axis = axis / np.sqrt((axis ** 2).sum())
    z_axis = np.array([0.0, 0.0, 1.0])
    tilt_axis = np.cross(z_axis, axis)
    if np.sqrt((tilt_axis ** 2).sum()) < 1e-5:
        tilt_axis = z_axis
    return np.asarray(utils.rotation_matrix(tilt_axis, np.arccos(np.dot(axis, z_axis))))


def generate_tube_xr(resolution, rot_axis, phi_start, phi_end, orbit_radius, std, r_isco, fov, std_clip=np.inf, normalize=True):
    """Arc of Gaussian blobs along the orbit between the angles ``phi_start`` and ``phi_end`` in steps of 0.015 rad
    (emission.py:62-117), as a ``utils.Volume``."""
    if orbit_radius < r_isco:
        raise AttributeError('hotspot center ({}) is is within r_isco: {}'.format(orbit_radius, r_isco))
    tilt = _tilt_to_axis(rot_axis)
    total, vol = 0.0, None
    for phi in np.arange(phi_start, phi_end, 0.015):
        center = np.matmul(tilt, [orbit_radius * np.cos(phi), orbit_radius * np.sin(phi), 0.0])
        vol = utils.gaussian_xr(resolution, center, std, fov=fov, std_clip=std_clip)
        total = total + vol.data
    if vol is None:
        raise AttributeError('empty angle range [{}, {})'.format(phi_start, phi_end))
    out = utils.Volume(total, vol.coords, vol.dims, dict(fov=fov, std=vol.attrs['std'], std_clip=std_clip))
    if normalize:
        out = utils.Volume(out.data / out.integrate(['x', 'y', 'z']), out.coords, out.dims, out.attrs)
    out.attrs.update(rot_axis=rot_axis, phi_start=phi_start, phi_end=phi_end)
    return out


def propogate_flatspace_emission(emission_0, Omega_3D, t_frames, t_start_obs=None, rot_axis=[0, 0, 1], M=None):
    """3-D movie of an initial emission sheared by the velocity field on its own grid -- no ray tracing, no light
    travel time (emission.py:305-341): warp of the grid points, then trilinear sampling of ``emission_0``."""
    x, y, z = np.meshgrid(emission_0.x, emission_0.y, emission_0.z, indexing='ij')
    t0 = (t_frames[0] if units.is_quantity(t_frames) else np.atleast_1d(t_frames)[0]) if t_start_obs is None else t_start_obs
    kw = {} if M is None else {'M': M}
    warped = velocity_warp_coords(coords=[x, y, z], Omega=Omega_3D, t_frames=t_frames, t_start_obs=t0, t_geos=0,
                                  t_injection=0, rot_axis=rot_axis, **kw)
    return interpolate_coords(emission_0, warped)


def normalize_stokes(movie, I_flux, P_flux, V_flux=None):
    """Scale a Stokes movie (nt, S, H, W) in place to a mean total flux ``I_flux`` and a mean linearly polarised flux
    ``P_flux`` of the light curves (emission.py:387-393)."""
    dolp = np.sqrt(np.sum(movie[:, 1:].sum(axis=(-1, -2)) ** 2, axis=1)).mean()
    movie[:, 0] *= I_flux / movie[:, 0].sum(axis=(-1, -2)).mean()
    movie[:, 1:3] *= P_flux / dolp
    if V_flux is not None:
        movie[:, 3] *= V_flux / movie[:, 3].sum(axis=(-1, -2)).mean()
    return movie
