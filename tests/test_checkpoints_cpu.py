"""flax msgpack checkpoint format (bhnerf_amd/checkpoints.py) -- pure host logic.

Known answers are built by hand from the published format of flax.serialization (0.3.4): ExtType(1) =
packb((shape, dtype.name, C-order bytes)), ExtType(3) for NumPy scalars, ExtType(2) for Python complex, dict keys
are strings, tuples become {'0': ..., '1': ...}."""
import collections
import os

import msgpack
import numpy as np
import pytest

from bhnerf_amd import checkpoints as ck


def test_known_answer_bytes_of_a_small_state_dict():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    enc = ck.msgpack_serialize({'step': np.asarray(7, dtype=np.int32), 'params': {'kernel': a}})
    # fixmap(2) { "step": ext8(type 1){ [[], "int32", bin(07 00 00 00)] }, "params": fixmap(1) {"kernel": ext8(1){...}} }
    step_payload = b'\x93' + b'\x90' + b'\xa5int32' + b'\xc4\x04' + np.int32(7).tobytes()
    arr_payload = b'\x93' + b'\x92\x02\x03' + b'\xa7float32' + b'\xc4\x18' + a.tobytes()
    want = (b'\x82' + b'\xa4step' + b'\xc7' + bytes([len(step_payload)]) + b'\x01' + step_payload +
            b'\xa6params' + b'\x81' + b'\xa6kernel' + b'\xc7' + bytes([len(arr_payload)]) + b'\x01' + arr_payload)
    assert enc == want
    back = ck.msgpack_restore(enc)
    assert back['step'].shape == () and back['step'].dtype == np.int32 and int(back['step']) == 7
    assert back['params']['kernel'].dtype == np.float32 and np.array_equal(back['params']['kernel'], a)


def test_roundtrip_types_tuples_and_namedtuples():
    Adam = collections.namedtuple('ScaleByAdamState', 'count mu nu')
    Sched = collections.namedtuple('ScaleByScheduleState', 'count')
    rng = np.random.default_rng(0)
    p = {'MLP_0': {'Dense_0': {'kernel': rng.standard_normal((21, 8)).astype(np.float32), 'bias': np.zeros(8, np.float32)}}}
    state = {'step': np.int32(3), 'params': p, 'opt_state': (Adam(np.asarray(3, np.int32), p, p), Sched(np.asarray(3, np.int32))),
             'z': 1.5 - 2j, 'vis': rng.standard_normal(5).astype(np.complex64), 'n': 12, 'name': 'x', 'f64': np.float64(0.25)}
    back = ck.msgpack_restore(ck.msgpack_serialize(state))
    assert set(back['opt_state']) == {'0', '1'} and set(back['opt_state']['0']) == {'count', 'mu', 'nu'}
    assert set(back['opt_state']['1']) == {'count'}
    assert np.array_equal(back['opt_state']['0']['mu']['MLP_0']['Dense_0']['kernel'], p['MLP_0']['Dense_0']['kernel'])
    assert back['z'] == 1.5 - 2j and back['vis'].dtype == np.complex64 and np.array_equal(back['vis'], state['vis'])
    assert back['n'] == 12 and back['name'] == 'x'
    assert isinstance(back['step'], np.generic) and back['step'] == 3 and back['f64'] == 0.25     # ExtType 3: NumPy scalars stay scalars
    with pytest.raises(ValueError):
        ck.msgpack_serialize({'o': np.array([object()])})


def test_save_restore_latest_keep_and_overwrite(tmp_path):
    d = str(tmp_path / 'run')
    assert ck.latest_checkpoint(d) is None and ck.restore_checkpoint(d, None) is None
    for step in (10, 20, 30):
        ck.save_checkpoint(d, {'step': np.asarray(step, np.int32), 'w': np.full(4, step, np.float32)}, step, keep=2)
    assert sorted(os.listdir(d)) == ['checkpoint_20', 'checkpoint_30']
    assert ck.latest_checkpoint(d).endswith('checkpoint_30')
    assert int(ck.restore_checkpoint(d, None)['step']) == 30
    assert int(ck.restore_checkpoint(d, None, step=20)['w'][0]) == 20
    assert int(ck.restore_checkpoint(os.path.join(d, 'checkpoint_20'), None)['step']) == 20
    with pytest.raises(ValueError):
        ck.save_checkpoint(d, {'step': np.asarray(5, np.int32)}, 5)             # older than the newest
    with pytest.raises(ValueError):
        ck.restore_checkpoint(d, None, step=999)
    ck.save_checkpoint(d, {'step': np.asarray(25, np.int32)}, 25, keep=2, overwrite=True)   # rewinds: newer files go
    assert sorted(os.listdir(d)) == ['checkpoint_20', 'checkpoint_25']

    class Target:
        def from_state_dict(self, sd):
            self.step = int(sd['step'])
            return self
    assert ck.restore_checkpoint(d, Target()).step == 25
    empty = Target()
    assert ck.restore_checkpoint(str(tmp_path / 'nothing'), empty) is empty


def test_reads_a_file_laid_out_like_the_reference_train_state(tmp_path):
    """What flax writes for TrainState.create(apply_fn, params, tx=optax.adam(schedule)) (network.py:171-182):
    built here with plain msgpack calls, independent of checkpoints.msgpack_serialize."""
    def nd(a):
        a = np.asarray(a)
        return msgpack.ExtType(1, msgpack.packb((a.shape, a.dtype.name, a.tobytes()), use_bin_type=True))
    k = np.linspace(-1, 1, 6, dtype=np.float32).reshape(3, 2)
    tree = lambda s: {'MLP_0': {'Dense_0': {'bias': nd(np.zeros(2, np.float32) + s), 'kernel': nd(k * s)}}}
    raw = msgpack.packb({'step': nd(np.asarray(41, np.int32)), 'params': tree(1.0),
                         'opt_state': {'0': {'count': nd(np.asarray(41, np.int32)), 'mu': tree(0.5), 'nu': tree(0.25)},
                                       '1': {'count': nd(np.asarray(41, np.int32))}}}, strict_types=True)
    d = tmp_path / 'ref'
    d.mkdir()
    (d / 'checkpoint_41').write_bytes(raw)
    sd = ck.restore_checkpoint(str(d), None)
    assert int(sd['step']) == 41 and int(sd['opt_state']['1']['count']) == 41
    assert np.allclose(sd['params']['MLP_0']['Dense_0']['kernel'], k)
    assert np.allclose(sd['opt_state']['0']['nu']['MLP_0']['Dense_0']['kernel'], 0.25 * k)
    assert ck.msgpack_serialize(sd) == raw            # and we write the same bytes back
