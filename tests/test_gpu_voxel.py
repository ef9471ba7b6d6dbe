"""GPU parity of the voxel forward renderer (emission.image_plane_dynamics / interpolate_coords)
against golden vectors produced by the reference's own functions (tests/golden/make_golden.py, G8)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / np.abs(b).max())


def test_interpolate_coords_golden(golden):
    from bhnerf_amd import emission
    g = golden('g8_dynamics')
    fov = float(g['axis'][-1] - g['axis'][0])
    out = emission.interpolate_coords((g['volume'], fov), g['points'])
    assert out.shape == g['interp'].shape and relerr(out, g['interp']) < 2e-6
    assert (g['interp'] == 0).any() and (g['interp'] > 0).any()             # inside and outside the grid
    dev_out = emission.interpolate_coords((g['volume'], [fov] * 3), torch.tensor(g['points'], device='cuda'))
    assert relerr(dev_out.cpu().numpy(), g['interp']) < 2e-6
    # edges: exactly on the last node is inside, one ulp outside is 0, NaN is 0 (scipy 'constant' mode)
    edge = np.array([[fov / 2, fov / 2, fov / 2], [fov / 2 * (1 + 1e-6), 0, 0], [np.nan, 0, 0], [-fov / 2, -fov / 2, -fov / 2]])
    vals = emission.interpolate_coords((g['volume'], fov), edge)
    assert vals[0] == pytest.approx(g['volume'][-1, -1, -1], rel=1e-6) and vals[1] == 0 and vals[2] == 0
    assert vals[3] == pytest.approx(g['volume'][0, 0, 0], rel=1e-6)


def test_image_plane_dynamics_golden(golden):
    from bhnerf_amd import emission, units
    g = golden('g8_dynamics')
    geos = types.SimpleNamespace(x=g['coords'][0], y=g['coords'][1], z=g['coords'][2], t=g['t_geos'], dtau=g['dtau'], Sigma=g['Sigma'])

    class Vol:                                      # xarray-like: dims + coordinate lookup + data
        dims = ('x', 'y', 'z')
        data = g['volume']
        def __getitem__(self, k):
            return g['axis']
    t = g['t_frames'] * units.hr
    img = emission.image_plane_dynamics(Vol(), geos, g['Omega'], t, float(g['t_injection']), J=1.0, doppler=False)
    assert img.shape == g['images'].shape and relerr(img, g['images']) < 1e-5
    imgJ = emission.image_plane_dynamics(Vol(), geos, g['Omega'], t, float(g['t_injection']), J=g['J'], doppler=False)
    assert imgJ.shape == g['images_J'].shape and relerr(imgJ, g['images_J']) < 1e-5
    fast = emission.image_plane_dynamics((g['volume'], float(g['axis'][-1] - g['axis'][0])), geos, g['Omega'], t,
                                         float(g['t_injection']), J=1.0, slow_light=False, doppler=False, t_start_obs=0.1 * units.hr)
    assert relerr(fast, g['images_fast']) < 1e-5
    movie4d = np.stack([g['volume']] * 3)            # one grid per frame (emission.py:289-293)
    img4 = emission.image_plane_dynamics((movie4d, float(g['axis'][-1] - g['axis'][0])), geos, g['Omega'], t,
                                         float(g['t_injection']), J=1.0, doppler=False)
    assert relerr(img4, g['images']) < 1e-5
    with pytest.raises(AttributeError):
        emission.image_plane_dynamics(Vol(), geos, g['Omega'], t, 0.0, doppler=True)       # needs geos.g
    with pytest.raises(AttributeError):
        emission.image_plane_dynamics(Vol(), geos, g['Omega'], t, 0.0, doppler=False, rot_axis=[1, 0, 0])


def test_config1_tutorial1_forward_render_at_size():
    """BASELINE config 1 (Tutorial1 forward render: spin 0 hotspot, 64 x 64 image, 32 samples per ray, 1 frame) at its
    own size: Kerr geodesics from the own tracer, a 64^3 Gaussian hotspot, Keplerian Omega, Doppler factor from the traced
    wave vectors -- the fused voxel renderer against the float64 oracle composition (pinned to the reference's output in
    tests/test_oracle_golden.py::test_image_plane_dynamics_golden).  Then 16 frames of the orbit: flux varies, the
    frame at t = 0 is the single-frame render."""
    from bhnerf_amd import constants, emission, kgeo, units
    from oracle import oracle_np as onp
    fov_M = 16.0
    geos = kgeo.image_plane_geos(0.0, np.deg2rad(60.0), (-fov_M / 2, fov_M / 2), (-fov_M / 2, fov_M / 2), ngeo=32, num_alpha=64, num_beta=64)
    rr = np.sqrt(geos.x ** 2 + geos.y ** 2)
    Omega = np.sign(1.0) * np.sqrt(geos.M) / (np.maximum(rr, 2.0) ** 1.5 + geos.spin * np.sqrt(geos.M))      # Keplerian, capped inside
    vol = emission.generate_hotspot_xr(resolution=(64, 64, 64), rot_axis=[0, 0, 1], rot_angle=0.0, orbit_radius=6.5,
                                       std=0.8, r_isco=constants.isco_pro(0.0), fov=(fov_M, 'GM/c^2'))
    arr, fov = emission._grid_of(vol)
    assert arr.shape == (64, 64, 64) and arr.max() > 0
    g = np.asarray(kgeo.doppler_factor(geos, kgeo.azimuthal_velocity_vector(geos, Omega)), dtype=np.float64)
    r32 = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)      # the device path holds the geometry in float32
    coords = r32(np.array([geos.x, geos.y, geos.z]))
    t_inj = -float(geos.r_o)
    for dop in (False, True):
        img = emission.image_plane_dynamics(vol, geos, Omega, 0.0 * units.hr, t_inj, J=1.0, doppler=dop)
        want = onp.image_plane_dynamics(arr, float(fov[0]), coords, r32(Omega), [0.0], t_inj, r32(geos.t), r32(g) if dop else 1.0, r32(geos.dtau), r32(geos.Sigma))[0]
        assert img.shape == (64, 64) and want.max() > 0
        assert relerr(img, want) < 2e-5, (dop, relerr(img, want))
    t = np.linspace(0.0, 1.5, 16) * units.hr
    movie = emission.image_plane_dynamics(vol, geos, Omega, t, t_inj, J=1.0, doppler=True)
    want = onp.image_plane_dynamics(arr, float(fov[0]), coords, r32(Omega), t.value if hasattr(t, 'value') else np.asarray(t), t_inj, r32(geos.t), r32(g), r32(geos.dtau), r32(geos.Sigma))
    assert movie.shape == (16, 64, 64) and relerr(movie, want) < 2e-5
    flux = movie.sum(axis=(1, 2))
    assert flux.std() > 0.02 * flux.mean()                           # the hotspot orbits: Doppler boosting modulates the flux


def test_polarised_flux_tube_movie_at_size():
    """The polarised case of the reference's light-curve notebooks at its own size (64 x 64 rays x 100 samples, Stokes I/Q/U
    emission factors from alma.image_plane_model, slow light, 5 frames): fused voxel renderer against the oracle composition."""
    from bhnerf_amd import alma, emission, units
    from oracle import oracle_np as onp
    params = dict(fov_M=40.0, z_width=4, rmin='ISCO', Q_frac=0.85, b_consts=dict(arad=0, avert=1, ator=0), Omega_dir='cw',
                  num_alpha=64, num_beta=64, t_start_obs=9.3)
    vol = emission.generate_tube_xr(resolution=(64, 64, 64), rot_axis=[0.0, 0.0, 1.0], phi_start=np.deg2rad(190), phi_end=np.deg2rad(270),
                                    orbit_radius=10.0, std=1.0, r_isco=6.0, fov=(40.0, 'GM/c^2'))
    arr, fov = emission._grid_of(vol)
    geos, Omega, J = alma.image_plane_model(np.deg2rad(12.0), 0.0, params)
    t = np.array([9.34056333, 9.35067, 9.36077667, 9.37088333, 9.38099])
    t_inj = -float(geos.r_o + 40.0 / 4)
    movie = emission.image_plane_dynamics(vol, geos, Omega, t * units.hr, t_inj, J, t_start_obs=9.3 * units.hr, doppler=False)
    r32 = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
    want = onp.image_plane_dynamics(arr.astype(np.float64), float(fov[0]), r32(np.array([geos.x, geos.y, geos.z])), r32(Omega), t, t_inj,
                                    r32(geos.t), 1.0, r32(geos.dtau), r32(geos.Sigma), J=r32(J), t_start_obs=9.3)
    assert movie.shape == (5, 3, 64, 64) and relerr(movie, want) < 5e-5, relerr(movie, want)
    lc, lw = movie.sum(axis=(-1, -2)), want.sum(axis=(-1, -2))
    assert np.abs(lc / lw - 1.0).max() < 1e-4

