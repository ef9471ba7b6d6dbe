"""flax-compatible training checkpoints.

The reference saves ``flax.training.checkpoints.save_checkpoint(dir, state, step, keep)`` files
(``optimization.py:118-121``) and reads them back with ``restore_checkpoint`` (``network.py:185``, ``842-848``,
``896-906``).  flax (pinned 0.3.4, ``requirements.txt:13``) is a third-party dependency that is not vendored in the
reference tree, so this module restates its published on-disk format (``flax/serialization.py``):

* the file ``<dir>/checkpoint_<step>`` is ONE msgpack object, the *state dict* of the train state;
* dict keys are strings; tuples / lists / namedtuples become dicts (``'0', '1', ...`` resp. the field names);
* every array is ``ExtType(1, packb((shape, dtype.name, C-order bytes)))``, NumPy scalars are ``ExtType(3, <same>)``,
  Python complex numbers ``ExtType(2, packb((re, im)))``;
* a ``TrainState`` built by ``train_state.TrainState.create(apply_fn, params, tx=optax.adam(schedule))``
  (``network.py:171-182``) serialises to ``{'step', 'params', 'opt_state': {'0': {'count','mu','nu'}, '1': {'count'}}}``
  (``optax.adam`` = ``chain(scale_by_adam, scale_by_schedule)``; ``apply_fn`` / ``tx`` are not pytree fields).

Parity: unpinned against a real flax file (flax / jax are not installable here; no checkpoint fixture ships with the
reference) -- the byte layout is checked against hand-built known answers of the published format
(``tests/test_checkpoints_cpu.py``).  Files written by round-1 builds of this package (``torch.save``) are still read.
"""
import os
import re

import msgpack
import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3


def _ndarray_bytes(arr):
    arr = np.asarray(arr)
    if arr.dtype.hasobject or arr.dtype.isalignedstruct:
        raise ValueError('object and structured arrays cannot be serialised')
    return msgpack.packb((tuple(arr.shape), arr.dtype.name, arr.tobytes('C')), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_bytes(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_bytes(x))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
    return x


def _ext_unpack(code, data):
    if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        arr = np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()
        return arr[()] if code == _EXT_NPSCALAR else arr
    if code == _EXT_COMPLEX:
        re_, im_ = msgpack.unpackb(data)
        return complex(re_, im_)
    return msgpack.ExtType(code, data)


def to_state_dict(target):
    """Nested structure -> nested dict with string keys (flax ``serialization.to_state_dict``)."""
    if hasattr(target, 'to_state_dict'):
        return target.to_state_dict()
    if isinstance(target, dict):
        return {str(k): to_state_dict(v) for k, v in target.items()}
    if isinstance(target, tuple) and hasattr(target, '_fields'):
        return {k: to_state_dict(getattr(target, k)) for k in target._fields}
    if isinstance(target, (list, tuple)):
        return {str(i): to_state_dict(v) for i, v in enumerate(target)}
    if hasattr(target, 'detach') and hasattr(target, 'cpu'):         # torch tensor
        return target.detach().cpu().numpy()
    return target


def msgpack_serialize(state_dict):
    """State dict -> bytes (flax ``serialization.msgpack_serialize``)."""
    return msgpack.packb(to_state_dict(state_dict), default=_ext_pack, strict_types=True)


def msgpack_restore(encoded):
    """bytes -> state dict of NumPy arrays (flax ``serialization.msgpack_restore``)."""
    return msgpack.unpackb(encoded, ext_hook=_ext_unpack, raw=False, strict_map_key=False)


def _steps_in(ckpt_dir, prefix):
    out = []
    if ckpt_dir and os.path.isdir(ckpt_dir):
        for f in os.listdir(ckpt_dir):
            m = re.fullmatch(re.escape(prefix) + r'(\d+)', f)
            if m:
                out.append((int(m.group(1)), os.path.join(ckpt_dir, f)))
    return sorted(out)


def latest_checkpoint(ckpt_dir, prefix='checkpoint_'):
    """Path of the checkpoint with the highest step, or None (flax ``checkpoints.latest_checkpoint``)."""
    found = _steps_in(ckpt_dir, prefix)
    return found[-1][1] if found else None


def save_checkpoint(ckpt_dir, target, step, prefix='checkpoint_', keep=1, overwrite=False):
    """Write ``<ckpt_dir>/<prefix><step>`` atomically and keep the ``keep`` newest (flax ``save_checkpoint``)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    found = _steps_in(ckpt_dir, prefix)
    if found and found[-1][0] >= int(step) and not overwrite:
        raise ValueError('a checkpoint with step %d >= %d exists in %s (pass overwrite=True)' % (found[-1][0], int(step), ckpt_dir))
    path = os.path.join(ckpt_dir, '%s%d' % (prefix, int(step)))
    tmp = path + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(msgpack_serialize(target))
    os.replace(tmp, path)
    found = _steps_in(ckpt_dir, prefix)
    if overwrite:
        for s, p in found:
            if s > int(step):
                os.remove(p)
        found = _steps_in(ckpt_dir, prefix)
    for _, stale in found[:-int(keep)] if keep > 0 else []:
        os.remove(stale)
    return path


def _read(path):
    with open(path, 'rb') as f:
        head = f.read(4)
        f.seek(0)
        if head[:2] == b'PK' or head[:1] == b'\x80':               # torch.save (zip / legacy pickle): round-1 files
            import torch
            sd = torch.load(f, map_location='cpu', weights_only=True)      # tensors only: never unpickle code
            return {'_legacy': True, **sd}
        return msgpack_restore(f.read())


def restore_checkpoint(ckpt_dir, target=None, step=None, prefix='checkpoint_'):
    """flax ``restore_checkpoint``: the newest (or the given) checkpoint of ``ckpt_dir`` restored into ``target``
    (anything with ``from_state_dict``), or returned as a plain state dict when ``target`` is None.  ``ckpt_dir`` may
    also be a checkpoint file.  No checkpoint found: ``target`` is returned unchanged."""
    if ckpt_dir and os.path.isfile(ckpt_dir):
        path = ckpt_dir
    elif step is not None:
        path = os.path.join(ckpt_dir, '%s%d' % (prefix, int(step)))
        if not os.path.exists(path):
            raise ValueError('no checkpoint %s' % path)
    else:
        path = latest_checkpoint(ckpt_dir, prefix)
    if path is None:
        return target
    sd = _read(path)
    if target is None:
        return sd
    if hasattr(target, 'from_state_dict'):
        return target.from_state_dict(sd)
    raise TypeError('cannot restore into a %s' % type(target).__name__)
