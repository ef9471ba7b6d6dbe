"""CPU tests of the host-side mirror of the reference interface (no GPU): the NumPy paths of the
stand-alone helpers against the golden vectors from the reference's own code, units, batching,
sharding, chunking of total_movie_loss, the TrainStep contract."""
import numpy as np
import pytest
import torch

from bhnerf_amd import constants, emission, kgeo, network, optimization, units, utils


def close(a, b, rtol=1e-12, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def test_constants_and_units(golden):
    g = golden('g0_constants')
    assert constants.GM_c3('hr') == float(g['GM_c3_hr']) == constants.GM_c3(units.hr)
    assert constants.GM_c3(None) == 1.0
    close(constants.GM_c3('s'), float(g['GM_c3_hr']) * 3600.0)
    close(constants.isco_pro(0.94), float(g['isco94']))
    q = np.linspace(0, 1, 5) * units.hr
    assert len(q) == 5 and units.unit_name(q.unit) == 'hr' and float(np.asarray(q.to('min').value)[1]) == 15.0
    with pytest.raises(AttributeError):
        constants.GM_c3('fortnight')


def test_standalone_helpers_numpy_path_golden(golden):
    g = golden('g1_warp')
    out = emission.velocity_warp_coords(g['coords'], g['Omega'], g['t_frames'], 0.0, g['t_geos'], float(g['t_injection']),
                                        t_units=units.hr)
    close(out, g['out_units'], atol=1e-12)
    out = emission.velocity_warp_coords(g['coords'], g['Omega'], g['t_frames'] * units.hr, 0.0 * units.hr, g['t_geos'],
                                        float(g['t_injection']))
    close(out, g['out_units'], atol=1e-12)
    close(emission.velocity_warp_coords(g['coords'], g['Omega'], g['t_frames'], 0.1, g['t_geos'], float(g['t_injection'])),
          g['out_nounits'], atol=1e-12)
    close(emission.velocity_warp_coords(g['coords'], 0.05, g['t_frames'], 0.0, g['t_geos'], float(g['t_injection']),
                                        t_units=units.hr), g['out_scalar_omega'], atol=1e-12)
    close(emission.velocity_warp_coords(g['coords'], g['Omega'], 0.4, 0.0, 0.0, 0.0), g['out_scalar_t'], atol=1e-12)
    close(utils.rotation_matrix([0, 0, 1], g['rot_angles']), g['rot'], atol=1e-15)
    # torch tensors go through the same expressions
    tw = emission.velocity_warp_coords(torch.tensor(g['coords']), torch.tensor(g['Omega']), g['t_frames'], 0.0,
                                       torch.tensor(g['t_geos']), float(g['t_injection']), t_units=units.hr)
    close(tw.numpy(), g['out_units'], atol=1e-11)
    g = golden('g3_fill')
    out = emission.fill_unsupervised_emission(g['emission'], g['coords'], float(g['rmin']), float(g['rmax']), float(g['z_width']))
    assert np.array_equal(out, g['out'])
    g = golden('g4_rt')
    close(kgeo.radiative_trasfer(g['emission'], g['g'], g['dtau'], g['Sigma']), g['out_arrays'])
    close(kgeo.radiative_transfer(g['emission'][0, 0], 1.3, 1.0, 0.5), g['out_scalars'])
    g = golden('g2_posenc')
    for d in (0, 1, 3, 5):
        close(network.posenc(g['x'], d), g['deg%d' % d], atol=1e-15)
    close(network.posenc(torch.tensor(g['x']), 3).numpy(), g['deg3'], atol=1e-13)
    assert utils.expand_dims(np.zeros((2, 3)), 4, axis=0).shape == (1, 1, 2, 3)
    assert utils.expand_dims(np.zeros((2, 3)), 4, axis=-1).shape == (2, 3, 1, 1)[:0] + utils.expand_dims(np.zeros((2, 3)), 4, -1).shape


def test_mlp_descriptor_matches_reference_layer_rule():
    assert network.MLP(4, 256).layer_dims(21) == [(21, 256), (256, 256), (256, 256), (277, 256), (256, 1)]
    assert [d[0] for d in network.MLP(8, 128).layer_dims(21)] == [21, 128, 128, 128, 128, 149, 128, 128, 128]
    tree = network.MLP(4, 64).init(seed=1, in_features=21)
    k = tree['MLP_0']['Dense_3']['kernel']
    assert tuple(k.shape) == (85, 64) and float(k.abs().max()) <= np.sqrt(6.0 / 85) and float(tree['MLP_0']['Dense_3']['bias'].abs().max()) == 0
    assert torch.equal(k, network.MLP(4, 64).init(seed=1, in_features=21)['MLP_0']['Dense_3']['kernel'])
    with pytest.raises(AttributeError):
        network.MLP(out_channel=3)


def test_temporal_batched_args_and_shard():
    t = np.linspace(0, 2, 12) * units.hr
    target = np.arange(12 * 6, dtype=np.float32).reshape(12, 2, 3)
    a = optimization.TemporalBatchedArgs(t, [target, np.ones_like(target)])
    b = optimization.TemporalBatchedArgs(t, [target, np.ones_like(target)])
    assert units.unit_name(a.t_units) == 'hr' and a.num_frames == 12
    for _ in range(3):      # identical draws on every rank (same seed, same call sequence), no replacement
        ia, ib = a.sample(6), b.sample(6)
        assert np.array_equal(ia, ib) and len(set(ia.tolist())) == 6
    tgt, sig, tf = a[np.array([3, 1, 7])]
    assert np.array_equal(np.asarray(tgt), target[[3, 1, 7]]) and np.allclose(tf, np.asarray(t.value)[[3, 1, 7]])
    assert optimization.device_count() == 1
    assert np.array_equal(optimization.shard(np.arange(8)), np.arange(8))
    with pytest.raises(AssertionError):
        optimization.TemporalBatchedArgs(t, [target[:5]])


def test_train_step_contract_and_total_movie_loss_chunking():
    t = np.linspace(0, 1, 7) * units.hr
    target = np.zeros((7, 2, 2), dtype=np.float32)
    calls = []

    def fake_step(state, t_units, dtype, tgt, sig, off, tf, *rest):
        calls.append(np.asarray(tf).copy())
        n = len(tf)
        return torch.full((1,), float(n)), state, torch.zeros((1, n, 2, 2))

    step = optimization.TrainStep('full', optimization.TemporalBatchedArgs(t, [target, target, target]), fake_step, fake_step, 1.0)
    rt = {'coords': None, 'Omega': None, 'J': 1.0, 'g': None, 'dtau': None, 'Sigma': None, 't_start_obs': 0.0,
          't_geos': None, 't_injection': 0.0}
    loss, frames = optimization.total_movie_loss(3, 'state', step, rt, return_frames=True)
    assert [len(c) for c in calls] == [3, 3, 1]                       # optimization.py:42-46 chunking incl. remainder
    assert loss == pytest.approx(7 / 7) and frames.shape == (7, 2, 2)
    both = step + step
    assert both.num_losses == 2
    with pytest.raises(AttributeError, match='only hr units supported'):
        optimization.TrainStep('full', optimization.TemporalBatchedArgs(np.linspace(0, 1, 7) * units.s, [target]), fake_step, fake_step, 1.0)
    with pytest.raises(ImportError, match='ehtim'):          # external package; eht_arrays is the array-based entry
        optimization.TrainStep.eht(t, None, 1e-10, 4, lambda *a, **k: None)
    A = (np.ones((7, 5, 4)) + 1j * np.ones((7, 5, 4)))
    eht = optimization.TrainStep.eht_arrays(t, np.zeros((7, 5), dtype=complex), np.ones((7, 5)), A, dtype='vis')
    tgt, sig, Ab, tf = eht.args[0][np.array([0, 2])]
    assert np.asarray(Ab).dtype == np.complex64 and np.asarray(tgt).dtype == np.complex64 and eht.dtype[0] == 'vis'
    fired = []
    log = optimization.LogFn(lambda opt: fired.append(opt.step), log_period=5)
    for s in (1, 2, 5, 7, 10):
        log(type('O', (), {'step': s})())
    assert fired == [1, 5, 10]


def test_raytracing_args_order_and_errors():
    geos = dict(x=np.zeros((2, 2, 3)), y=np.zeros((2, 2, 3)), z=np.zeros((2, 2, 3)), dtau=np.ones((2, 2, 3)),
                Sigma=np.ones((2, 2, 3)), t=np.zeros((2, 2, 3)), g=np.ones((2, 2, 3)))
    rt = network.raytracing_args(geos, np.ones((2, 2, 3)), -1000.0, 0.0 * units.hr)
    assert list(rt) == ['coords', 'Omega', 'J', 'g', 'dtau', 'Sigma', 't_start_obs', 't_geos', 't_injection']   # network.py:882-892
    assert rt['coords'].shape == (3, 2, 2, 3) and rt['coords'].dtype == np.float32 and rt['J'] == 1.0
    del geos['g']
    with pytest.raises(AttributeError):
        network.raytracing_args(geos, 1.0, 0.0, 0.0)


def test_schedule_and_checkpoint_helpers(tmp_path):
    st = network.TrainState.__new__(network.TrainState)
    st.step, st.num_iters, st.lr_init, st.lr_final = 0, 10, 1e-3, 1e-5
    assert st.learning_rate(0) == 1e-3 and st.learning_rate(10) == 1e-5 and st.learning_rate(25) == 1e-5
    assert st.learning_rate(5) == pytest.approx((1e-3 - 1e-5) * 0.5 + 1e-5)
    assert network.latest_checkpoint('') is None and network.latest_checkpoint(str(tmp_path)) is None
    for n in (5, 50, 7):
        (tmp_path / ('checkpoint_%d' % n)).write_bytes(b'x')
    assert network.latest_checkpoint(str(tmp_path)).endswith('checkpoint_50')
    pred = network.NeRF_Predictor(20.0, 6.0, 20.0, 4.0, net_width=128)
    pred.save_params(str(tmp_path))
    again = network.NeRF_Predictor.from_yml(str(tmp_path))
    assert (again.scale, again.rmin, again.rmax, again.z_width, again.net_width, again.posenc_deg) == (20.0, 6.0, 20.0, 4.0, 128, 3)


def test_quantity_arithmetic_and_loss_readable_by_numpy():
    """What the reference's fit scripts do with times and with ``opt.loss`` (scripts/Fit_ALMA_LP_Apr11_SgrA_Flare.py:66-69, 98)."""
    import torch
    from bhnerf_amd import optimization
    t = np.array([9.4, 10.2, 11.5]) * units.hr
    split = 9.33 * units.hr + 103.0 * units.min
    assert np.array_equal(np.asarray(t <= split), [True, True, False]) and np.array_equal(np.asarray(t > split), [False, False, True])
    assert len(t[np.asarray(t <= split)]) == 2
    assert float((t[1] - t[0]).to('min').value) == pytest.approx(48.0)
    loss = torch.tensor([10.0, 1000.0]).as_subclass(optimization.HostReadable)
    assert float(np.log10(np.mean(loss))) == pytest.approx(np.log10(505.0))
    assert np.asarray(loss).tolist() == [10.0, 1000.0] and float(torch.as_tensor(loss).mean()) == 505.0


def test_gaussian_volume_and_hotspot_generator():
    """utils.gaussian_xr / emission.generate_hotspot_xr (utils.py:48-95, emission.py:10-60) on the xarray-free Volume."""
    from bhnerf_amd import emission, utils as U
    vol = U.gaussian_xr([33, 33, 33], (1.0, -2.0, 0.5), 1.5, fov=(16.0, 'GM/c^2'))
    assert vol.dims == ('x', 'y', 'z') and vol.shape == (33, 33, 33) and np.allclose(vol.x, np.linspace(-8, 8, 33))
    i, j, k = np.unravel_index(np.argmax(vol.data), vol.shape)
    assert (vol.x[i], vol.y[j], vol.z[k]) == (1.0, -2.0, 0.5) and vol.data.max() == 1.0
    assert np.isclose(vol.data[i + 3, j, k], np.exp(-0.5 * (1.5 / 1.5) ** 2))                   # 3 cells = 1.5 M = 1 std
    clipped = U.gaussian_xr([33, 33, 33], (0.0, 0.0, 0.0), 1.0, fov=(16.0, 'GM/c^2'), std_clip=2.0)
    assert clipped.data.min() == 0.0 and (clipped.data[clipped.data > 0] > np.exp(-2.0)).all()
    img = U.gaussian_xr([9, 17], (0.25, -0.25), (0.1, 0.2))
    assert img.dims == ('y', 'x') and img.shape == (17, 9)                      # resolution = (nx, ny)
    with pytest.raises(AttributeError):
        U.gaussian_xr([9, 9, 9], (0.0, 0.0), 1.0)
    # hotspot on the orbit, rotated about the axis; unit volume integral
    hs = emission.generate_hotspot_xr([65, 65, 65], [0, 0, 1], np.pi / 2, 5.0, 0.8, 3.0, (20.0, 'GM/c^2'))
    assert np.isclose(hs.integrate(['x', 'y', 'z']), 1.0)
    i, j, k = np.unravel_index(np.argmax(hs.data), hs.shape)
    assert abs(hs.x[i]) < 0.2 and abs(hs.y[j] - 5.0) < 0.2 and abs(hs.z[k]) < 0.2
    tilted = emission.generate_hotspot_xr([65, 65, 65], [1, 0, 0], 0.0, 5.0, 0.8, 3.0, (20.0, 'GM/c^2'), normalize=False)
    c = tilted.attrs['center']
    assert np.isclose(np.dot(c, [1, 0, 0]), 0.0, atol=1e-12) and np.isclose(np.linalg.norm(c), 5.0)   # orbit plane is normal to the axis
    with pytest.raises(AttributeError):
        emission.generate_hotspot_xr([9, 9, 9], [0, 0, 1], 0.0, 2.0, 0.5, 3.0, (20.0, 'M'))
    # the volume is accepted where the reference takes a DataArray
    arr, fov = emission._grid_of(hs)
    assert arr.shape == (65, 65, 65) and fov == [20.0, 20.0, 20.0]


def test_lds_swizzle_of_the_fused_width128_backward_is_conflict_free():
    """csrc/fused_bwd128.hip keeps 128-point x 128-feature images in LDS as 256-byte rows of sixteen 16-byte chunks at
    off(row, ch) = 256 row + 16 (ch ^ swz(row)).  Three access patterns touch them; MI355X_MICROARCH.md (LDS) gives the lane
    groups that share an LDS cycle and the bank of a byte address.  The swizzle must be conflict-free for all three:
      * ds_write_b128 / row stores of a finished tile: 8 consecutive lanes (= 8 consecutive rows, one chunk position) per
        cycle, bank = (addr / 4) mod 32  ->  chunk mod 8 distinct over 8 aligned rows;
      * ds_read_b128 / row reads: lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, bank = (addr / 4) mod 64  ->  the
        sixteen chunks distinct;
      * ds_read_b64_tr_b16 / transposed reads: 32 lanes = 4 aligned rows x 4 consecutive chunks x two 8-byte halves, bank =
        (addr / 4) mod 64  ->  chunk >> 2 distinct over 4 aligned rows.
    (Round 4's swizzle met the last two only: 21 % of the LDS-active cycles of bwd128_kernel were bank conflicts.)"""
    swz = lambda r: ((r & 3) << 2) | (((r >> 2) & 3) ^ (r & 2))            # the formula of fused_bwd128.hip (rowb, tr_first / tr_second, voffH)
    for c in range(16):                                                      # any chunk position (and lane half: one more xor)
        for g in range(16):
            assert len({((c ^ swz(8 * g + i)) & 7) for i in range(8)}) == 8
        for grp in (list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))):
            assert len({(c ^ swz(p)) & 15 for p in grp}) == 16
    for a in range(32):
        assert len({swz(4 * a + b) >> 2 for b in range(4)}) == 4
    old = lambda r: ((r & 3) << 2) | ((r >> 2) & 3)
    assert len({old(i) & 7 for i in range(8)}) == 4                          # round 4: every bank hit twice by a row store


def test_ray_span_of_a_compacted_layout():
    """engine.ray_span (bhn_geom.ray_span, ABI 4): the most 32-point groups the consecutive points of one ray lie in."""
    import torch
    from bhnerf_amd import engine
    t = lambda v: torch.tensor(v, dtype=torch.int32)
    assert engine.ray_span(t([])) == 0
    assert engine.ray_span(t([5] * 32)) == 1                                  # one ray filling one group exactly
    assert engine.ray_span(t([0] * 31 + [1] * 2)) == 2                        # the second ray straddles the boundary
    assert engine.ray_span(t([0] * 10 + [3] * 22 + [4] * 32 + [9] * 1)) == 1  # runs that end on group boundaries
    assert engine.ray_span(t([0] * 31 + [1] * 34)) == 3                       # points 31 .. 64: groups 0, 1 and 2
    assert engine.ray_span(t([7] * 100)) == 4
    rng = torch.Generator().manual_seed(3)
    counts = torch.randint(1, 23, (500,), generator=rng)                      # config-3-like: at most 22 in-domain samples per ray
    ray = torch.repeat_interleave(torch.arange(500, dtype=torch.int32), counts)
    assert engine.ray_span(ray) == 2


def test_bench_roofline_bookkeeping():
    """bench.py's per-kernel figures: flops each MLP kernel EXECUTES of the algorithm, tape bytes from the
    library's own layout (bhn_tape_info, host-only), the clock from the kernels' stamps."""
    import ctypes as C
    import os
    import sys
    import __graft_entry__ as entry
    entry.build()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from bhnerf_amd import _hip
    lib = _hip.lib()

    def info(depth, width, mode):
        m = _hip.make_model(depth, width, 3, True, 8.0, 0.0, 8.0, 4.0)
        out = (C.c_int64 * _hip.BHN_TAPE_INFO_N)()
        assert lib.bhn_tape_info(C.byref(m), mode, 8192, out, _hip.BHN_TAPE_INFO_N) == 0
        return {'fwd_write': out[0], 'chain_write': out[1], 'chain_read': out[2], 'dw_read': out[3],
                'flags': {k: bool(out[4] & v) for k, v in _hip.TAPE_FLAGS.items()}}

    fwd, chain, dw, train = bench.mlp_flops(4, 256)
    assert (fwd, train) == (415232, 1234944) and fwd + chain + dw == train
    i256 = info(4, 256, _hip.BHN_BF16)
    k = bench.kernel_flops(4, 256, i256['flags'])
    # under ga0_chain the delta chain carries dW_0 (2 F W = 10,752) and the output layer's delta (2 W = 512); the dW kernel has no layer-0 job
    assert k == {bench.FWD_NAME: 415232, bench.CHAIN_NAME: 393216 + 512 + 10752, 'dw_kernel': 415232 - 10752}
    assert sum(k.values()) == train - 10752              # SURVEY's step figure also counts an input gradient of the skip features nobody computes
    kf = bench.kernel_flops(4, 256, info(4, 256, _hip.BHN_F32)['flags'])
    assert kf[bench.CHAIN_NAME] == 393216 + 512 and kf['dw_kernel'] == 415232
    i128 = info(4, 128, _hip.BHN_BF16)
    assert bench.kernel_flops(4, 128, i128['flags']) == {bench.FWD_NAME: 109312, bench.FUSED_NAME: 98304 + 256 + 109312}
    assert bench.tape_bytes_per_point(i256) == {bench.FWD_NAME: 1796.0, bench.CHAIN_NAME: 1224.0, 'dw_kernel': 2720.0}
    assert bench.tape_bytes_per_point(i128) == {bench.FWD_NAME: 596.0, bench.FUSED_NAME: 600.0}
    # four stamps per kernel slot {s_memtime, s_memrealtime} x {start, end}; s_memrealtime counts at 100 MHz
    stamps = np.zeros(16, dtype=np.int64)
    stamps[8:12] = [1000, 50, 1000 + 1_850_000, 50 + 100_000]                              # slot 2: 1.85e6 core cycles in 1 ms
    assert bench.clock_mhz(stamps, bench.CLK_SLOT[bench.CHAIN_NAME]) == 1850.0 and bench.clock_mhz(stamps, 0) is None
