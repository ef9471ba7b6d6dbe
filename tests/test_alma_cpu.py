"""Host-side ALMA helpers (bhnerf/alma.py) and emission.rotate_evpa (emission.py:395-407)."""
import numpy as np
import pytest

from bhnerf_amd import alma, emission, units

PARAMS = dict(num_alpha=6, num_beta=6, fov_M=16.0, z_width=4.0, rmin='ISCO', Q_frac=0.5,
              b_consts=dict(arad=0.0, avert=1.0, ator=0.0), Omega_dir='cw', t_start_obs=9.5)


def test_rotate_evpa_known_answers():
    qu = np.array([1.0, 0.0])
    assert np.allclose(emission.rotate_evpa(qu, np.pi / 4), [0.0, 1.0], atol=1e-15)          # EVPA +45 deg: Q -> U
    assert np.allclose(emission.rotate_evpa(qu, np.pi / 2), [-1.0, 0.0], atol=1e-15)         # +90 deg: Q -> -Q
    iquv = np.array([3.0, 1.0, 2.0, 0.5])
    out = emission.rotate_evpa(iquv, 0.3)
    p = np.exp(0.6j) * (1.0 + 2.0j)
    assert np.allclose(out, [3.0, p.real, p.imag, 0.5], atol=1e-15)
    assert np.allclose(emission.rotate_evpa(iquv[:3], 0.3), out[:3], atol=1e-15)
    with pytest.raises(AttributeError):
        emission.rotate_evpa(np.zeros(5), 0.1)


def test_rotate_evpa_axis_and_norm():
    rng = np.random.default_rng(0)
    s = rng.standard_normal((7, 3, 4))
    out = emission.rotate_evpa(s, -0.7, axis=1)
    assert out.shape == s.shape
    assert np.array_equal(out[:, 0], s[:, 0])
    assert np.allclose(out[:, 1] ** 2 + out[:, 2] ** 2, s[:, 1] ** 2 + s[:, 2] ** 2)          # |P| is invariant
    assert np.allclose(emission.rotate_evpa(out, 0.7, axis=1), s)
    # per-element rotation of the complex polarisation
    p = np.exp(-1.4j) * (s[:, 1] + 1j * s[:, 2])
    assert np.allclose(out[:, 1], p.real) and np.allclose(out[:, 2], p.imag)


@pytest.fixture(scope='module')
def model():
    return alma.image_plane_model(np.deg2rad(20.0), 0.3, PARAMS)


def test_image_plane_model(model):
    geos, Omega, J = model
    assert J.shape == (3, 6, 6, 100) and np.isfinite(J).all()
    assert Omega.shape == (6, 6, 100) and (Omega[geos.r > 0] < 0).all()                      # 'cw' rotates clockwise
    # linear polarisation fraction is bounded by Q_frac wherever there is emission
    I, P = J[0], np.hypot(J[1], J[2])
    assert (P <= PARAMS['Q_frac'] * np.abs(I) * (1 + 1e-9) + 1e-12).all()
    assert (I >= 0).all() and I.max() > 0


def test_image_plane_model_rotation_and_direction(model):
    geos, Omega, J = model
    _, Omega_ccw, J_rot = alma.image_plane_model(np.deg2rad(20.0), 0.3, dict(PARAMS, rmin=3.0, Omega_frac=0.8), rot_angle=0.4)
    # rmin only moves the normalisation domain, Omega_frac scales the angular velocity
    assert np.allclose(Omega_ccw, 0.8 * Omega)
    assert J_rot.shape == J.shape
    _, Omega2, J2 = alma.image_plane_model(np.deg2rad(20.0), 0.3, PARAMS, rot_angle=0.4)
    assert np.allclose(J2, emission.rotate_evpa(J, 0.4), atol=1e-12)
    _, Omega3, _ = alma.image_plane_model(np.deg2rad(20.0), 0.3, dict(PARAMS, Omega_dir='ccw'))
    assert np.allclose(Omega3, -Omega)


def test_get_raytracing_args():
    rts = alma.get_raytracing_args(np.deg2rad(20.0), 0.3, PARAMS, stokes=['Q', 'U'])
    assert len(rts) == 1
    rt = rts[0]
    assert list(rt) == ['coords', 'Omega', 'J', 'g', 'dtau', 'Sigma', 't_start_obs', 't_geos', 't_injection']
    assert rt['J'].shape == (2, 6, 6, 100) and rt['coords'].shape == (3, 6, 6, 100)
    assert rt['t_injection'] == -(1000.0 + 4.0)
    assert units.unit_name(rt['t_start_obs'].unit) in ('hr', 'h') and float(rt['t_start_obs'].value) == 9.5
    assert np.isfinite(rt['g']).all() and rt['g'].max() > 0
    np.random.seed(0)
    jittered = alma.get_raytracing_args(np.deg2rad(20.0), 0.3, PARAMS, num_subpixel_rays=2)
    assert len(jittered) == 2 and not np.allclose(jittered[0]['coords'], jittered[1]['coords'])


def test_preprocess_data(tmp_path):
    pd = pytest.importorskip('pandas')
    rng = np.random.default_rng(1)
    # two scans separated by a gap; 4 s cadence
    t = np.concatenate([9.0 + np.arange(400) * 4 / 3600.0, 9.0 + 0.6 + np.arange(400) * 4 / 3600.0])
    df = pd.DataFrame({'time': t, 'Q': rng.standard_normal(t.size), 'U': rng.standard_normal(t.size)})
    path = tmp_path / 'lc.csv'
    df.to_csv(path)
    target, t_frames = alma.preprocess_data(str(path), 10, I_hs_mean=0.3, P_sha=0.1, chi_sha=20.0, de_rot_angle=-15.0,
                                            t_start=9.1, t_end=9.9)
    tv = np.asarray(t_frames.value)
    assert target.shape == (len(tv), 3) and np.all(target[:, 0] == 0.3)
    assert np.all(np.diff(tv) > 0) and tv.min() >= 9.1 and tv.max() <= 9.9
    # restate the windowing by hand: rolling(w).mean().loc[::w] keeps the windows that END at rows w, 2w, ... of the
    # selected period; a mean is dropped when it lies 160 s or more after the previous window mean
    sel = df[(df.time >= 9.1) & (df.time <= 9.9)].to_numpy()
    ends = np.arange(10, len(sel), 10)
    means = np.stack([sel[e - 9:e + 1].mean(axis=0) for e in ends])
    keep = np.concatenate([[True], np.diff(means[:, 0]) < 160.0 / 3600.0])
    means = means[keep]
    assert np.allclose(tv, means[:, 0])
    qu = means[:, 1:] - 0.1 * np.array([np.cos(np.deg2rad(40.0)), np.sin(np.deg2rad(40.0))])
    p = np.exp(2j * np.deg2rad(-15.0)) * (qu[:, 0] + 1j * qu[:, 1])
    assert np.allclose(target[:, 1], p.real) and np.allclose(target[:, 2], p.imag)
    assert not keep.all()                                       # the first window after the scan gap was dropped


def test_chi2_df_without_checkpoints(tmp_path):
    pytest.importorskip('pandas')
    df = alma.chi2_df([10.0, 20.0], 0.3, [0, 1], PARAMS, str(tmp_path / 'inc_{}_seed_{}'), np.zeros(3), np.zeros((3, 3)))
    assert df.index.name == 'inc' and list(df.columns) == ['seed 0', 'seed 1'] and df.shape == (2, 2)
    assert np.isnan(df.values).all()
    df = alma.chi2_df(10.0, [0.1, 0.3, 0.5], [0], PARAMS, str(tmp_path / 'spin_{}_seed_{}'), np.zeros(3), np.zeros((3, 3)))
    assert df.index.name == 'spin' and df.shape == (3, 1)
    with pytest.raises(AttributeError):
        alma.chi2_df([1.0, 2.0], [0.1, 0.2], [0], PARAMS, 'x{}{}', np.zeros(3), np.zeros((3, 3)))


def test_against_reference_golden(tmp_path):
    """tests/golden/g10_alma.npz: the reference's own rotate_evpa / preprocess_data outputs (make_golden.py G10)."""
    import os
    pd = pytest.importorskip('pandas')
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g10_alma.npz'))
    assert np.allclose(emission.rotate_evpa(gold['s2'], 0.37), gold['rot2'], rtol=1e-13, atol=1e-15)
    assert np.allclose(emission.rotate_evpa(gold['s3'], -1.2, axis=1), gold['rot3'], rtol=1e-13, atol=1e-15)
    assert np.allclose(emission.rotate_evpa(gold['s4'], 2.9, axis=2), gold['rot4'], rtol=1e-13, atol=1e-15)
    path = tmp_path / 'lc.csv'
    pd.DataFrame({'time': gold['lc_time'], 'I': gold['lc_I'], 'Q': gold['lc_Q'], 'U': gold['lc_U']}).to_csv(path)
    w, I_hs, P_sha, chi_sha, derot, t0, t1 = gold['pre_args']
    target, t_frames = alma.preprocess_data(str(path), int(w), I_hs, P_sha, chi_sha, derot, t_start=t0, t_end=t1)
    assert target.shape == gold['pre_target'].shape
    assert np.allclose(target, gold['pre_target'], rtol=1e-12, atol=1e-14)
    assert np.allclose(np.asarray(t_frames.to('hr').value), gold['pre_t_hr'], rtol=1e-14)


def test_logging_helpers_against_reference_golden(tmp_path):
    """utils.mse / psnr / intensity_to_nchw / world_to_image_coords (utils.py:9-11, 160-193) against fixture g10."""
    import json
    import os
    from bhnerf_amd import optimization, utils
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g10_alma.npz'))
    assert utils.mse(gold['vol'], gold['vol_est']) == pytest.approx(float(gold['mse']), rel=1e-14)
    assert utils.psnr(gold['vol'], gold['vol_est']) == pytest.approx(float(gold['psnr']), rel=1e-14)
    assert np.allclose(utils.world_to_image_coords(gold['wc'], (8.0, 8.0, 10.0), (16, 16, 20)), gold['wc_img'], rtol=1e-14)
    nchw = utils.intensity_to_nchw(gold['vol'])
    assert nchw.shape == gold['nchw'].shape == (6, 3, 5, 4) and np.allclose(nchw, gold['nchw'], atol=1e-12)
    assert np.allclose(utils.intensity_to_nchw(gold['vol'], 'magma', 1.0), gold['nchw_g1'], atol=1e-12)
    # file-backed writer (tensorboardX is not installed here)
    w = optimization.SummaryWriter(str(tmp_path / 'log'))
    w.add_scalar('datafit/x', 0.5, global_step=3)
    w.add_images('emission/estimate', nchw, global_step=3, dataformats='NCWH')
    w.close()
    if w._tb is None:
        rec = [json.loads(l) for l in open(tmp_path / 'log' / 'scalars.jsonl')]
        assert rec == [{'tag': 'datafit/x', 'value': 0.5, 'step': 3}]
        assert np.array_equal(np.load(tmp_path / 'log' / 'emission_estimate_3.npy'), nchw)


def test_normalize_stokes_tube_and_flatspace_propagation():
    """emission.normalize_stokes against the reference's output (fixture g10); generate_tube_xr and
    propogate_flatspace_emission (emission.py:62-117, 305-341) through their defining properties."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g10_alma.npz'))
    mv = gold['stokes_movie']
    assert np.allclose(emission.normalize_stokes(mv[:, :3].copy(), 2.5, 0.4), gold['stokes_norm3'], rtol=1e-13)
    out4 = emission.normalize_stokes(mv.copy(), 2.5, 0.4, V_flux=-0.1)
    assert np.allclose(out4, gold['stokes_norm4'], rtol=1e-13)
    assert np.isclose(out4[:, 0].sum(axis=(-1, -2)).mean(), 2.5) and np.isclose(out4[:, 3].sum(axis=(-1, -2)).mean(), -0.1)
    # tube: unit integral, all of it within one std-scale of the orbit circle, only between the two angles
    tube = emission.generate_tube_xr([49, 49, 49], [0, 0, 1], 0.0, np.pi / 2, 5.0, 0.6, 3.0, (16.0, 'M'))
    assert np.isclose(tube.integrate(['x', 'y', 'z']), 1.0)
    X, Y, Z = np.meshgrid(tube.x, tube.y, tube.z, indexing='ij')
    w = tube.data / tube.data.sum()
    assert abs((w * np.hypot(X, Y)).sum() - 5.0) < 0.15 and abs((w * Z).sum()) < 1e-12
    assert (w * ((X < -1.5) | (Y < -1.5))).sum() < 1e-3                                   # first quadrant only
    with pytest.raises(AttributeError):
        emission.generate_tube_xr([9, 9, 9], [0, 0, 1], 0.0, 1.0, 2.0, 0.5, 3.0, (16.0, 'M'))
    # flat-space propagation: rigid rotation (constant Omega > 0) by a quarter turn moves the hotspot from +x to +y ...
    hs = emission.generate_hotspot_xr([41, 41, 41], [0, 0, 1], 0.0, 5.0, 0.8, 3.0, (16.0, 'M'), normalize=False)
    quarter = np.pi / 2                                                                  # Omega * t with t in units of M
    mov = emission.propogate_flatspace_emission(hs, 1.0, np.array([0.0, quarter]))
    assert mov.shape == (2, 41, 41, 41) and np.allclose(mov[0], hs.data, atol=1e-6)
    i, j, k = np.unravel_index(np.argmax(mov[1]), mov[1].shape)
    assert abs(hs.x[i]) < 0.5 and abs(abs(hs.y[j]) - 5.0) < 0.5 and abs(hs.z[k]) < 0.5
    # ... (the emission seen at x is the initial one at R(-theta) x: the pattern turns counter-clockwise by +theta)
    assert hs.y[j] > 0
