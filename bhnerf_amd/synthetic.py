"""Synthetic "geodesic" inputs for tests and benchmarks.

The external Kerr ray tracer ``kgeo`` is not part of the reference tree, so the hot path is fed
straight-line rays through a flat-space volume with the same array shapes, dtypes and value ranges
as ``bhnerf.kgeo.image_plane_geos`` output: coords (3,H,W,G), Omega/t_geos/g/dtau/Sigma (H,W,G),
optional Stokes factors J (S,H,W,G).  Everything is generated in float64 and cast to float32.
"""
import numpy as np


def synthetic_geodesics(H, W, G, fov_M=16.0, inc_deg=60.0, spin=0.0, S=0, seed=0):
    rng = np.random.default_rng(seed)
    rmax = fov_M / 2.0
    alpha, beta = np.meshgrid(np.linspace(-rmax, rmax, H), np.linspace(-rmax, rmax, W), indexing='ij')
    length = 2.4 * rmax                                   # chord across the sphere r <= 1.2 rmax
    s = np.linspace(0.0, length, G)
    inc = np.deg2rad(inc_deg)
    # image-plane basis (observer at inclination `inc` from the z axis), rays along the line of sight
    ex = np.array([1.0, 0.0, 0.0])
    ey = np.array([0.0, np.cos(inc), -np.sin(inc)])
    los = np.array([0.0, np.sin(inc), np.cos(inc)])
    origin = alpha[..., None, None] * ex + beta[..., None, None] * ey          # (H,W,1,3)
    pts = origin + (s - 0.5 * length)[None, None, :, None] * los               # (H,W,G,3)
    coords = np.moveaxis(pts, -1, 0)
    r = np.sqrt((coords ** 2).sum(0))
    cos_th = coords[2] / np.maximum(r, 1e-9)
    Sigma = r ** 2 + spin ** 2 * cos_th ** 2 + 1e-3
    ds = length / (G - 1)
    out = {
        'coords': coords,
        'Omega': 1.0 / (np.maximum(r, 0.5) ** 1.5 + spin),
        't_geos': -(1000.0 - s) * np.ones_like(r),
        'g': rng.uniform(0.6, 1.4, r.shape),
        'dtau': ds / Sigma,                               # dtau * Sigma = path-length element
        'Sigma': Sigma,
    }
    if S:
        I = rng.uniform(0.5, 1.5, r.shape)
        chi = rng.uniform(0.0, np.pi, r.shape)
        out['J'] = np.stack([I, 0.85 * I * np.cos(2 * chi), 0.85 * I * np.sin(2 * chi)])[:S]
    else:
        out['J'] = 1.0
    out = {k: (np.ascontiguousarray(v, dtype=np.float32) if not np.isscalar(v) else v) for k, v in out.items()}
    out['t_injection'] = -(1000.0 + fov_M / 4.0)          # alma.py:78 convention: nothing is pre-injection
    return out


def hotspot_movie(geo, t_frames, GM_c3, orbit_radius=5.5, sigma=0.7, t_start_obs=0.0):
    """Target movie: a Gaussian hotspot (Tutorial3 cell 2 values) orbiting with the Keplerian Omega,
    rendered with the same warp + ray sum.  NumPy, one-off set-up cost (not on the timed path)."""
    x, y, z = (geo['coords'][i].astype(np.float64) for i in range(3))
    w = geo['g'].astype(np.float64) ** 2 * geo['dtau'] * geo['Sigma']
    frames = []
    for t in np.asarray(t_frames, dtype=np.float64):
        th = ((t - t_start_obs) / GM_c3 + geo['t_geos'] - geo['t_injection']) * geo['Omega']
        xw, yw = np.cos(th) * x + np.sin(th) * y, np.cos(th) * y - np.sin(th) * x
        e = np.exp(-((xw - orbit_radius) ** 2 + yw ** 2 + z ** 2) / (2 * sigma ** 2))
        frames.append((e * w).sum(-1))
    return np.stack(frames).astype(np.float32)
