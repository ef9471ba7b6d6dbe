"""NumPy restatement of the bhnerf hot path (test oracle, not shipped code).

dtype-generic: feed float64 arrays for the ground truth, float32 arrays to see the
reference's own single-precision noise floor.  Citations are to /root/reference.
"""
import numpy as np

# constants.py:13,17 -- G*M/c^3 for M = 4.154e6 M_sun, in hours (value measured by
# evaluating the reference's astropy expression in the build container).
GM_C3_SGRA_HR = 0.0056834692768060625
SAFE_SIN_PERIOD = 100.0 * np.pi        # network.py:16


def expand_dims(x, ndim, axis=0):
    """utils.py:215-219: insert unit axes at `axis` until x has `ndim` dims."""
    x = np.asarray(x)
    while x.ndim < ndim:
        x = np.expand_dims(x, axis=min(axis, x.ndim))
    return x


def rotation_matrix(axis, angle):
    """utils.py:97-132: Euler-Rodrigues rotation about `axis` by `angle` (array ok).

    Returns shape (3, 3, *angle.shape).
    """
    axis = np.asarray(axis, dtype=np.result_type(angle, np.float32))
    axis = axis / np.sqrt(np.dot(axis, axis))
    half = np.asarray(angle) / 2.0
    a = np.cos(half)
    b, c, d = (-axis[0] * np.sin(half), -axis[1] * np.sin(half), -axis[2] * np.sin(half))
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc]])


def velocity_warp_coords(coords, Omega, t_frames, t_start_obs, t_geos, t_injection,
                         rot_axis=(0, 0, 1), GM_c3=1.0):
    """emission.py:143-211.  `GM_c3` is 1.0 when t_units is None, else GM/c^3 in
    t_units (emission.py:183-185).  Output (*t_frames.shape, *spatial, 3); NaN where
    the point is seen before the injection time (emission.py:204-205)."""
    coords = np.asarray(coords)
    Omega = np.asarray(Omega)
    t_frames = np.asarray(t_frames)
    if Omega.ndim == 0:                                            # emission.py:192-193
        Omega = expand_dims(Omega, coords.ndim - 1, axis=-1)
    if t_frames.ndim != 0:                                         # emission.py:196-198
        coords = expand_dims(coords, coords.ndim + t_frames.ndim, 1)
        t_frames = expand_dims(t_frames, t_frames.ndim + Omega.ndim, -1)
    t_geos = (t_frames - t_start_obs) / GM_c3 + np.asarray(t_geos)  # emission.py:200
    t_M = t_geos - t_injection                                     # emission.py:201
    theta = np.asarray(t_M * Omega)
    theta = np.where(t_M < 0.0, np.full_like(theta, np.nan), theta)  # emission.py:204-205
    inv_rot = rotation_matrix(rot_axis, -theta)                    # emission.py:207
    warped = np.sum(inv_rot * coords, axis=1)                      # emission.py:209
    return np.moveaxis(warped, 0, -1)                              # emission.py:210


def fill_unsupervised_emission(emission, coords, rmin=0.0, rmax=np.inf, z_width=2.0,
                               fill_value=0.0):
    """emission.py:343-374: zero emission outside rmin<=r<=rmax, |z|<=z_width."""
    r_sq = np.sum(np.array([np.squeeze(c) ** 2 for c in coords]), axis=0)
    emission = np.where(r_sq < rmin ** 2, np.full_like(emission, fill_value), emission)
    emission = np.where(r_sq > rmax ** 2, np.full_like(emission, fill_value), emission)
    emission = np.where(np.abs(coords[2]) > z_width, np.full_like(emission, fill_value), emission)
    return emission


def radiative_trasfer(emission, g, dtau, Sigma):
    """kgeo.py:595-622 (name misspelt in the reference): sum_k g^2 e dtau Sigma."""
    nd = np.ndim(emission)
    g, dtau, Sigma = (expand_dims(v, nd) for v in (g, dtau, Sigma))
    return (g ** 2 * emission * dtau * Sigma).sum(axis=-1)


def safe_sin(x):
    """network.py:16 (floor-mod, so negative x is shifted up by 100*pi)."""
    return np.sin(x % SAFE_SIN_PERIOD)


def posenc(x, deg):
    """network.py:98-122: [x, sin(2^i x) (i-major, xyz-minor), sin(2^i x + pi/2)]."""
    if deg == 0:
        return x
    scales = np.array([2 ** i for i in range(deg)])
    xb = np.reshape(x[..., None, :] * scales[:, None], list(x.shape[:-1]) + [-1])
    four_feat = safe_sin(np.concatenate([xb, xb + 0.5 * np.pi], axis=-1))
    return np.concatenate([x, four_feat], axis=-1)


def mlp_layer_dims(net_depth, net_width, in_features, do_skip=True, out_channel=1):
    """Input width of every Dense layer implied by network.py:52-62."""
    dims = []
    cur = in_features
    skip_layer = net_depth // 2 if do_skip else None
    for i in range(net_depth):
        dims.append((cur, net_width))
        cur = net_width
        if do_skip and i % skip_layer == 0 and i > 0:
            cur = net_width + in_features
    dims.append((cur, out_channel))
    return dims


def mlp_apply(params, x, net_depth=4, do_skip=True):
    """network.py:49-62 with flax nn.Dense semantics (y = x @ kernel + bias, kernel
    (in,out)); params = {'Dense_i': {'kernel','bias'}} for i = 0..net_depth."""
    inputs = x
    skip_layer = net_depth // 2 if do_skip else None
    for i in range(net_depth):
        p = params['Dense_%d' % i]
        x = x @ p['kernel'] + p['bias']
        x = np.maximum(x, 0)                                       # nn.relu
        if do_skip and i % skip_layer == 0 and i > 0:
            x = np.concatenate([x, inputs], axis=-1)
    p = params['Dense_%d' % net_depth]
    return x @ p['kernel'] + p['bias']


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def predictor_apply(params, t_frames, coords, Omega, t_start_obs, t_geos, t_injection,
                    GM_c3=GM_C3_SGRA_HR, scale=1.0, rmin=0.0, rmax=np.inf, z_width=np.inf,
                    posenc_deg=3, net_depth=4, do_skip=True):
    """NeRF_Predictor.__call__, network.py:219-233.  `params` is the {'MLP_0': ...}
    tree.  Returns emission (*t_frames.shape, *spatial)."""
    warped = velocity_warp_coords(coords, Omega, t_frames, t_start_obs, t_geos, t_injection,
                                  GM_c3=GM_c3)
    valid = np.isfinite(warped)                                    # network.py:226
    net_in = np.where(valid, warped, np.zeros_like(warped))        # network.py:227
    out = mlp_apply(params['MLP_0'], posenc(net_in / scale, posenc_deg), net_depth, do_skip)
    emission = sigmoid(out[..., 0] - 10.0)                         # network.py:230
    emission = fill_unsupervised_emission(emission, coords, rmin, rmax, z_width)
    return np.where(valid[..., 0], emission, np.zeros_like(emission))  # network.py:232


def image_plane_prediction(emission, J, g, dtau, Sigma):
    """network.py:415-419 given the predictor's emission (incl. the jnp.squeeze quirk)."""
    if not np.isscalar(J):
        Jx = expand_dims(J, emission.ndim + 1, 0)
        emission = Jx * expand_dims(emission, emission.ndim + 1, 1)
        emission = np.squeeze(emission)                            # network.py:418
    return radiative_trasfer(emission, g, dtau, Sigma)


def loss_image(images, target, sigma, offset, scale, dtype):
    """network.py:476-484."""
    if dtype == 'full':
        loss = np.sum(np.abs((images - target - offset) / sigma) ** 2)
    elif dtype == 'lc':
        lightcurve = images.sum(axis=(-1, -2))
        loss = np.sum(np.abs((lightcurve - target - offset) / sigma) ** 2)
    else:
        raise AttributeError('image dtype ({}) not supported'.format(dtype))
    return scale * loss


def loss_eht(images, target, sigma, A, scale, dtype):
    """network.py:541-564."""
    vec = images.reshape(*images.shape[:-2], -1, 1)
    vec = expand_dims(vec, A.ndim, axis=-3)
    vis = np.squeeze(np.matmul(A, vec), -1)
    if dtype == 'vis':
        if vis.ndim != target.ndim:
            raise AttributeError('visibilities ndim mismatch')
        chisq = np.sum((np.abs(vis - target) / sigma) ** 2)
    elif dtype == 'amp':
        if vis.ndim != target.ndim:
            raise AttributeError('visibilities ndim mismatch')
        chisq = np.sum(np.abs((np.abs(vis) - target) / sigma) ** 2)
    elif dtype == 'cphase':
        if vis.ndim != target.ndim + 1:
            raise AttributeError('visibilities ndim mismatch')
        clphase = np.angle(np.prod(vis, axis=-2))
        chisq = np.sum((1.0 - np.cos(target - clphase)) / (sigma ** 2))
    else:
        raise AttributeError('eht dtype ({}) not supported'.format(dtype))
    return scale * chisq


def linear_lr(step, lr_init, lr_final, num_iters):
    """optax.polynomial_schedule(init, end, power=1, transition_steps=N), network.py:173:
    (init-end)*(1-min(t,N)/N)^1 + end, with t = number of updates already applied."""
    frac = 1.0 - min(step, num_iters) / float(num_iters)
    return (lr_init - lr_final) * frac + lr_final


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """optax.adam (scale_by_adam then scale by -lr), network.py:174.  `t` is the
    1-based count of this update.  Returns (p, m, v)."""
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    mhat = m / (1.0 - b1 ** t)
    vhat = v / (1.0 - b2 ** t)
    return p - lr * mhat / (np.sqrt(vhat) + eps), m, v


def he_uniform_params(rng, net_depth, net_width, in_features, do_skip=True, out_channel=1,
                      dtype=np.float32):
    """he_uniform kernels U(+-sqrt(6/fan_in)), zero bias (network.py:50; flax Dense
    defaults).  NOT bit-compatible with jax.random.PRNGKey -- tests inject weights."""
    tree = {}
    for i, (fi, fo) in enumerate(mlp_layer_dims(net_depth, net_width, in_features, do_skip,
                                                out_channel)):
        lim = np.sqrt(6.0 / fi)
        tree['Dense_%d' % i] = {
            'kernel': rng.uniform(-lim, lim, size=(fi, fo)).astype(dtype),
            'bias': np.zeros((fo,), dtype=dtype)}
    return {'MLP_0': tree}


def map_coordinates_linear(grid, index):
    """scipy / jax.scipy.ndimage.map_coordinates(grid, index, order=1, mode='constant', cval=0.0) restated:
    `index` is (3, ...) in voxel units; samples whose index lies outside [0, n-1] on any axis (or is NaN) are 0."""
    grid = np.asarray(grid)
    ix, iy, iz = (np.asarray(index[i], dtype=np.float64) for i in range(3))
    nx, ny, nz = grid.shape
    inside = (ix >= 0) & (ix <= nx - 1) & (iy >= 0) & (iy <= ny - 1) & (iz >= 0) & (iz <= nz - 1)
    ixc, iyc, izc = np.where(inside, ix, 0.0), np.where(inside, iy, 0.0), np.where(inside, iz, 0.0)
    x0 = np.minimum(np.floor(ixc).astype(np.int64), max(nx - 2, 0)); x1 = np.minimum(x0 + 1, nx - 1)
    y0 = np.minimum(np.floor(iyc).astype(np.int64), max(ny - 2, 0)); y1 = np.minimum(y0 + 1, ny - 1)
    z0 = np.minimum(np.floor(izc).astype(np.int64), max(nz - 2, 0)); z1 = np.minimum(z0 + 1, nz - 1)
    tx, ty, tz = ixc - x0, iyc - y0, izc - z0
    out = ((grid[x0, y0, z0] * (1 - tz) + grid[x0, y0, z1] * tz) * (1 - ty) + (grid[x0, y1, z0] * (1 - tz) + grid[x0, y1, z1] * tz) * ty) * (1 - tx) + \
          ((grid[x1, y0, z0] * (1 - tz) + grid[x1, y0, z1] * tz) * (1 - ty) + (grid[x1, y1, z0] * (1 - tz) + grid[x1, y1, z1] * tz) * ty) * tx
    return np.where(inside, out, 0.0)


def grid_predictor_apply(grid, t_frames, coords, Omega, t_start_obs, t_geos, t_injection, GM_c3=GM_C3_SGRA_HR, scale=1.0,
                         rmin=0.0, rmax=np.inf, z_width=np.inf):
    """GRID_Predictor.__call__ (network.py:306-353): warp -> where(valid, warped, 0) -> voxel index
    (x + scale)/(2 scale) (res - 1) -> trilinear sample (0 outside the grid) -> sigmoid(. - 10) -> domain fill ->
    0 where the warped x is not finite."""
    grid = np.asarray(grid)
    res = grid.shape[0]
    warped = velocity_warp_coords(coords, Omega, t_frames, t_start_obs, t_geos, t_injection, GM_c3=GM_c3)
    valid = np.isfinite(warped)
    net_in = np.moveaxis(np.where(valid, warped, np.zeros_like(warped)), -1, 0)
    index = (net_in + scale) / (2.0 * scale) * (res - 1.0)
    emission = sigmoid(map_coordinates_linear(grid, index) - 10.0)
    emission = fill_unsupervised_emission(emission, coords, rmin, rmax, z_width)
    return np.where(valid[..., 0], emission, np.zeros_like(emission))


def image_plane_dynamics(volume, fov, coords, Omega, t_frames, t_injection, t_geos, g, dtau, Sigma, J=1.0,
                         t_start_obs=None, slow_light=True, GM_c3=GM_C3_SGRA_HR):
    """emission.py:235-303 restated from the pieces above (float64): velocity warp (slow light: with t_geos, else 0,
    emission.py:269) -> trilinear sampling of `volume` (nx,ny,nz) spanning [-fov/2, fov/2]^3 (emission.py:213-233) -> 0
    where the warp is NaN (before the injection) -> x J -> radiative transfer.  `t_start_obs` defaults to t_frames[0]
    (emission.py:274).  Returns (nt, [S], H, W).  Pinned to the reference's own output in fixture g8
    (tests/test_oracle_golden.py)."""
    volume = np.asarray(volume, dtype=np.float64)
    t_frames = np.atleast_1d(np.asarray(t_frames, dtype=np.float64))
    t0 = t_frames[0] if t_start_obs is None else float(t_start_obs)
    tg = np.asarray(t_geos, dtype=np.float64) if slow_light else 0.0
    warped = velocity_warp_coords(np.asarray(coords, dtype=np.float64), np.asarray(Omega, dtype=np.float64), t_frames, t0, tg,
                                  float(t_injection), GM_c3=GM_c3)                    # (nt, *spatial, 3)
    n = volume.shape
    index = [(warped[..., i] + fov / 2.0) / fov * (n[i] - 1) for i in range(3)]
    em = map_coordinates_linear(volume, index)
    em = np.where(np.isnan(warped).any(axis=-1), 0.0, em)
    if np.ndim(J) > 0:
        em = np.asarray(J, dtype=np.float64)[None] * em[:, None]                     # (nt, S, *spatial)
    return radiative_trasfer(em, g, dtau, Sigma)

