"""bhnerf.network hot-path API on the MI355X engine (reference: bhnerf/network.py).

Same names, argument order and error behaviour as the reference for the functions on the hot
path: ``posenc``, ``MLP``, ``NeRF_Predictor``, ``image_plane_prediction``, ``loss_fn_image``,
``gradient_step_image``, ``test_image``, ``sample_3d_grid``, ``raytracing_args``.  Where the
reference builds a JAX expression that XLA compiles, this module calls the fused HIP kernels through
``engine`` (C ABI, include/bhnerf_hip.h).  Parameters/outputs are torch device tensors.

Data-parallel semantics (network.py:620): each process (one per GPU) computes the SUM of
chi^2 over its own frames, gradients are AVERAGED over processes with one RCCL all-reduce of the
flat gradient buffer, then every rank applies the identical Adam update.
"""
import glob
import os
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch
import yaml

from . import _hip, checkpoints, constants, engine, units, utils

safe_sin = lambda x: (torch.sin(torch.remainder(x, 100 * np.pi)) if isinstance(x, torch.Tensor)
                      else np.sin(x % (100 * np.pi)))                       # network.py:16


def posenc(x, deg):
    """[x, sin(2^i x), sin(2^i x + pi/2)] (network.py:98-122); NumPy or torch, same feature order
    as the fused kernel's prologue."""
    if deg == 0:
        return x
    xp = utils._xp(x)
    if xp is torch:
        scales = torch.tensor([2.0 ** i for i in range(deg)], dtype=x.dtype, device=x.device)
        xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], -1)
        return torch.cat([x, safe_sin(torch.cat([xb, xb + 0.5 * np.pi], dim=-1))], dim=-1)
    scales = np.array([2 ** i for i in range(deg)])
    xb = np.reshape(x[..., None, :] * scales[:, None], list(x.shape[:-1]) + [-1])
    return np.concatenate([x, safe_sin(np.concatenate([xb, xb + 0.5 * np.pi], axis=-1))], axis=-1)


class ParamTree(dict):
    """{'MLP_0': {'Dense_i': {'kernel','bias'}}} whose leaves are views of one flat f32 buffer."""
    flat = None


class MLP:
    """Shape descriptor of the reference MLP (network.py:18-64).  The evaluation itself only exists
    fused inside ``NeRF_Predictor.apply`` (csrc/fused_fwd.hip)."""

    def __init__(self, net_depth=4, net_width=128, activation='relu', out_channel=1, do_skip=True):
        if out_channel != 1:
            raise AttributeError('out_channel={} not supported (every reference driver uses 1)'.format(out_channel))
        self.net_depth, self.net_width, self.activation = net_depth, net_width, activation
        self.out_channel, self.do_skip = out_channel, do_skip

    def layer_dims(self, in_features):
        dims, cur = [], in_features
        skip_layer = self.net_depth // 2 if self.do_skip else None
        for i in range(self.net_depth):
            dims.append((cur, self.net_width))
            cur = self.net_width
            if self.do_skip and i % skip_layer == 0 and i > 0:        # network.py:59-61
                cur = self.net_width + in_features
        dims.append((cur, self.out_channel))
        return dims

    def init(self, seed, in_features, device='cpu'):
        """he_uniform kernels, zero biases (network.py:50 / flax Dense defaults).  torch RNG: NOT
        bit-identical to jax.random.PRNGKey(seed) (parity unpinned for initialisation)."""
        gen = torch.Generator().manual_seed(int(seed))
        tree = {}
        for i, (fi, fo) in enumerate(self.layer_dims(in_features)):
            lim = float(np.sqrt(6.0 / fi))
            k = (torch.rand((fi, fo), generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim
            tree['Dense_%d' % i] = {'kernel': k.to(device), 'bias': torch.zeros((fo,), dtype=torch.float32, device=device)}
        return {'MLP_0': tree}


def _default_device():
    if not torch.cuda.is_available():
        raise _hip.HipError('no HIP device visible: bhnerf_amd has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


class TrainState:
    """flax TrainState stand-in: step, params, Adam moments, apply_fn (network.py:171-189)."""

    def __init__(self, apply_fn, flat, predictor, num_iters, lr_init, lr_final):
        self.apply_fn, self.predictor = apply_fn, predictor
        self.flat = flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.step = 0
        self.num_iters, self.lr_init, self.lr_final = int(num_iters), float(lr_init), float(lr_final)
        self.grad = torch.zeros(flat.numel() + _world()[1], dtype=torch.float32, device=flat.device)
        # Opt-in (north_star: "all-reduce ... overlapped with the next forward"): the all-reduce of step k runs
        # under the forward and backward of step k+1 and its gradient is applied after them -- one-step-stale gradients,
        # NOT the reference's synchronous pmean (network.py:620), hence off by default.
        self.overlap_allreduce = False
        self._grad2 = None
        self._pending = None                    # (work handle, buffer) of the all-reduce in flight

    def grad_buffer(self):
        """Flat gradient + one loss slot per rank (the single all-reduce message); re-sized when the process group was
        initialised after this state was built.  With an all-reduce in flight: the other buffer of the pair."""
        want = self.flat.numel() + _world()[1]
        if self.grad.numel() != want:
            self.finish_allreduce()
            self.grad = torch.zeros(want, dtype=torch.float32, device=self.flat.device)
            self._grad2 = None
        if self._pending is not None and self._pending[1] is self.grad:
            if self._grad2 is None:
                self._grad2 = torch.zeros_like(self.grad)
            return self._grad2
        return self.grad

    def exchange_overlapped(self, buf, n, loss, rank, world):
        """Stale-gradient data parallelism (overlap_allreduce): start this step's all-reduce asynchronously, then complete
        the PREVIOUS step's (it ran under this step's forward and backward) and apply its gradient.  Returns the loss
        vector of the completed step (first step: this rank's own loss in every slot; no update yet)."""
        import torch.distributed as dist
        prev = self._pending
        buf[n:].zero_()
        buf[n + rank] = loss.reshape(-1)[0]
        self._pending = (dist.all_reduce(buf, async_op=True), buf)
        if prev is None:                        # no completed step yet: this rank's own loss in every slot (a NaN filler
            return loss.reshape(-1)[:1].expand(world).clone()      # would reach Optimizer.loss, the log functions and every mean)
        prev[0].wait()                          # the compute stream waits for the collective; the host does not block
        self.apply_gradients(prev[1][:n], grad_scale=1.0 / world)
        return prev[1][n:].clone()

    def finish_allreduce(self):
        """Complete and apply the all-reduce still in flight (end of a run, before a checkpoint is written)."""
        if self._pending is not None:
            work, buf = self._pending
            self._pending = None
            work.wait()
            self.apply_gradients(buf[:self.flat.numel()], grad_scale=1.0 / _world()[1])
        return self

    @property
    def params(self):
        tree = ParamTree(self.predictor.engine().unflatten(self.flat))
        tree.flat = self.flat
        return tree

    def learning_rate(self, count=None):
        """optax.polynomial_schedule(lr_init, lr_final, power=1, transition_steps=num_iters)."""
        count = self.step if count is None else count
        frac = 1.0 - min(count, self.num_iters) / float(self.num_iters)
        return (self.lr_init - self.lr_final) * frac + self.lr_final

    def apply_gradients(self, grads, grad_scale=1.0):
        engine.adam_step(self.flat, grads, self.m, self.v, self.step + 1, self.learning_rate(), grad_scale=grad_scale)
        self.step += 1
        return self

    def to_state_dict(self):
        """The state dict flax writes for TrainState.create(apply_fn, params, tx=optax.adam(schedule))
        (network.py:171-182; checkpoints.py): step, params tree, Adam moments mu / nu as params-shaped trees."""
        eng = self.predictor.engine()
        host = lambda t: {k: host(v) for k, v in t.items()} if isinstance(t, dict) else t.cpu().numpy().copy()
        tree = lambda flat: host(eng.unflatten(flat))
        count = np.asarray(self.step, dtype=np.int32)
        return {'step': count, 'params': tree(self.flat),
                'opt_state': {'0': {'count': count, 'mu': tree(self.m), 'nu': tree(self.v)}, '1': {'count': count}}}

    def from_state_dict(self, sd):
        """Restore from a flax state dict (or from the torch.save dict of round-1 builds of this package)."""
        if sd.get('_legacy'):
            self.step = int(sd['step'])
            self.flat.copy_(sd['params']); self.m.copy_(sd['m']); self.v.copy_(sd['v'])
            return self
        eng = self.predictor.engine()
        self.flat.copy_(eng.flatten(sd['params']))
        opt = sd.get('opt_state') or {}
        adam = opt.get('0', opt) if isinstance(opt, dict) else {}
        if 'mu' in adam and 'nu' in adam:
            self.m.copy_(eng.flatten(adam['mu'])); self.v.copy_(eng.flatten(adam['nu']))
        else:                                    # parameters only (e.g. a hand-written file): fresh moments
            self.m.zero_(); self.v.zero_()
        self.step = int(np.asarray(sd.get('step', adam.get('count', 0))))
        return self

    def state_dict(self):
        return self.to_state_dict()

    def load_state_dict(self, sd):
        return self.from_state_dict(sd)


def _fingerprint(v):
    """Cache key of one ray-tracing argument (NeRF_Predictor.geometry): identity + what an in-place edit would change.
    BEST EFFORT for NumPy arrays (a strided 257-element sample: an edit that touches none of the sampled elements is not
    seen -- pass a new array or call clear_geometry_cache()); exact for torch tensors (in-place version counter)."""
    if v is None:
        return v
    if np.isscalar(v):
        return ('nan',) if (isinstance(v, float) and v != v) else v      # (a NaN never compares equal: it would rebuild the geometry every step)
    if isinstance(v, torch.Tensor):
        return (id(v), tuple(v.shape), v._version)
    if isinstance(v, (list, tuple)):
        return tuple(_fingerprint(x) for x in v)
    a = v if type(v) is np.ndarray else np.asarray(v)
    if a.size == 0:
        return (id(v), a.shape, a.dtype.str)
    step = max(1, (a.size - 1) // 256)                              # 257 samples from the first element to (nearly) the last
    flat = a.reshape(-1) if a.flags.c_contiguous else a.flat       # (a view: ~2.5 us per array and step)
    return (id(v), a.shape, a.dtype.str, hash(flat[::step].tobytes()))


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _world():
    import torch.distributed as dist
    if _dist_on():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class NeRF_Predictor:
    """Predict emission at (time-frame, point) samples (network.py:124-252).  Field order matches
    the reference dataclass so that ``NeRF_Predictor(rmax, rmin, rmax, z_width)`` works unchanged.
    Extra keyword-only fields: ``mode`` ('bf16' throughput / 'f32' parity) and ``device``."""

    def __init__(self, scale=1.0, rmin=0.0, rmax=np.inf, z_width=np.inf, posenc_deg=3, posenc_var=2e-5,
                 net_depth=4, net_width=128, activation='relu', out_channel=1, do_skip=True, *, mode='bf16',
                 device=None):
        if activation not in ('relu', torch.relu, torch.nn.functional.relu):
            raise AttributeError('only the relu activation is supported')
        if out_channel != 1:
            raise AttributeError('out_channel must be 1')
        self.scale, self.rmin, self.rmax, self.z_width = scale, rmin, rmax, z_width
        self.posenc_deg, self.posenc_var = posenc_deg, posenc_var
        self.net_depth, self.net_width, self.activation = net_depth, net_width, 'relu'
        self.out_channel, self.do_skip = out_channel, do_skip
        self.mode, self.device = mode, device
        self._engine, self._engine_key, self._geoms = None, None, OrderedDict()

    # -- engine / geometry caches ----------------------------------------------------------------
    def engine(self):
        key = (self.scale, self.rmin, self.rmax, self.z_width, self.posenc_deg, self.net_depth, self.net_width,
               self.do_skip, self.mode)
        if self._engine is None or self._engine_key != key:
            dev = torch.device(self.device) if self.device is not None else _default_device()
            self._engine = engine.FusedPredictor(self.net_depth, self.net_width, self.posenc_deg, self.do_skip,
                                                 self.scale, self.rmin, self.rmax, self.z_width, self.mode, dev)
            self._engine_key = key
            self._geoms.clear()
        return self._engine

    def geometry(self, coords, Omega, t_geos, J=None, g=None, dtau=None, Sigma=None):
        """Prepared RayGeometry for these arrays, cached on the identity of the inputs AND a fingerprint of their
        contents (the reference's ray-tracing arguments are immutable jax arrays; NumPy arrays and tensors are not):
        torch tensors by their in-place version counter (exact), NumPy arrays by shape, dtype and a strided sample of
        ~257 elements -- an in-place edit that touches none of the sampled elements is not seen: pass a new array, or
        call ``clear_geometry_cache()``."""
        key = tuple(_fingerprint(v) for v in (coords, Omega, t_geos, J, g, dtau, Sigma)) + (self.rmin, self.rmax, self.z_width)
        hit = self._geoms.get(key)
        if hit is not None:
            self._geoms.move_to_end(key)            # LRU: logging geometries must not evict the training ray sets
            return hit[0]
        dev = self.engine().device
        one = 1.0
        geom = engine.RayGeometry(coords, Omega, one if g is None else g, one if dtau is None else dtau,
                                  one if Sigma is None else Sigma, t_geos, J, self.rmin, self.rmax, self.z_width, dev)
        self._geoms[key] = (geom, (coords, Omega, t_geos, J, g, dtau, Sigma))   # keep ids alive
        while len(self._geoms) > 32:            # > the sub-pixel ray sets of a run (scripts use up to 10): a training step
            self._geoms.popitem(last=False)     # picks one at random, so a smaller cache would rebuild geometry every step
        return geom

    def clear_geometry_cache(self):
        self._geoms.clear()
        self._graph_epoch = getattr(self, '_graph_epoch', 0) + 1      # captured training steps (optimization.GraphedImageStep) are dropped too

    # -- reference API -----------------------------------------------------------------------------
    def init_params(self, raytracing_args=None, seed=1):
        mlp = MLP(self.net_depth, self.net_width, self.activation, self.out_channel, self.do_skip)
        eng = self.engine()
        flat = eng.flatten(mlp.init(seed, 3 + 6 * self.posenc_deg, 'cpu'))
        tree = ParamTree(eng.unflatten(flat))
        tree.flat = flat
        return tree

    def flat_params(self, params):
        flat = getattr(params, 'flat', None)
        if flat is None:
            flat = self.engine().flatten(params['params'] if 'params' in params else params)
        return flat

    def init_state(self, params, num_iters=5000, lr_init=1e-4, lr_final=1e-6, lr_inject=None, checkpoint_dir=''):
        # lr_inject is inert in the reference too: the t_injection parameter is commented out
        # (network.py:176-180, 235).
        flat = self.flat_params(params).clone()
        state = TrainState(self.apply, flat, self, num_iters, lr_init, lr_final)
        if checkpoint_dir:                                                     # network.py:185
            state = checkpoints.restore_checkpoint(checkpoint_dir, state)
        return state

    def apply(self, variables, t_frames, t_units, coords, Omega, t_start_obs, t_geos, t_injection):
        """emission = sigmoid(MLP(posenc(warp(coords)/scale)) - 10), masked (network.py:191-237).
        Returns a float32 device tensor shaped (*t_frames.shape, *coords.shape[1:])."""
        eng = self.engine()
        params = variables['params'] if isinstance(variables, dict) and 'params' in variables else variables
        eng.pack(self.flat_params(params))
        geom = self.geometry(coords, Omega, t_geos)
        tM0, scalar_t = _frame_offsets(t_frames, t_units, t_start_obs, t_injection, eng.device)
        e = eng.predict(geom, tM0)
        sp = tuple(np.shape(coords)[1:])
        return e.reshape(sp) if scalar_t else e.reshape((tM0.numel(),) + sp)

    __call__ = apply

    def save_params(self, directory, filename='NeRF_Predictor_params.yml'):
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        keys = ['scale', 'rmin', 'rmax', 'z_width', 'posenc_deg', 'posenc_var', 'net_depth', 'net_width',
                'out_channel', 'do_skip']                                       # network.py:242
        with open(directory.joinpath(filename), 'w') as f:
            yaml.dump({k: (float(getattr(self, k)) if isinstance(getattr(self, k), (float, np.floating))
                           else getattr(self, k)) for k in keys}, f)

    @classmethod
    def from_yml(cls, directory, filename='NeRF_Predictor_params.yml', **kw):
        params = yaml.safe_load(Path(directory).joinpath(filename).read_text())
        return cls(**params, **kw)


class GRID_Predictor(NeRF_Predictor):
    """Emission as a learnable voxel grid (network.py:254-370): fields ``scale, rmin, rmax, z_width, grid_res``;
    parameters ``{'grid': (res,res,res)}`` initialised to -10 (network.py:334).  Same driver interface as
    NeRF_Predictor (``init_params / init_state / apply / save_params / from_yml``); kernels ``bhn_grid_*``."""

    def __init__(self, scale=1.0, rmin=0.0, rmax=np.inf, z_width=np.inf, grid_res=64, *, device=None, mode='f32'):
        self.scale, self.rmin, self.rmax, self.z_width, self.grid_res = scale, rmin, rmax, z_width, int(grid_res)
        self.device, self.mode = device, 'f32'
        self._engine, self._engine_key, self._geoms = None, None, OrderedDict()

    def engine(self):
        key = (self.scale, self.grid_res)
        if self._engine is None or self._engine_key != key:
            dev = torch.device(self.device) if self.device is not None else _default_device()
            self._engine = engine.GridEngine(self.grid_res, self.scale, dev)
            self._engine_key = key
            self._geoms.clear()
        return self._engine

    def init_params(self, raytracing_args=None, seed=1):
        eng = self.engine()
        flat = torch.full((eng.nparams,), -10.0, dtype=torch.float32, device=eng.device)
        tree = ParamTree(eng.unflatten(flat))
        tree.flat = flat
        return tree

    def save_params(self, directory, filename='GRID_Predictor_params.yml'):
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        # (the reference's key list is the NeRF one, network.py:359, and so writes only these four; grid_res is added so
        #  that from_yml can rebuild the predictor)
        with open(directory.joinpath(filename), 'w') as f:
            yaml.dump({k: (float(getattr(self, k)) if isinstance(getattr(self, k), (float, np.floating)) else getattr(self, k))
                       for k in ('scale', 'rmin', 'rmax', 'z_width', 'grid_res')}, f)

    @classmethod
    def from_yml(cls, directory, filename='GRID_Predictor_params.yml', **kw):
        params = yaml.safe_load(Path(directory).joinpath(filename).read_text())
        return cls(**params, **kw)


def latest_checkpoint(checkpoint_dir):
    return checkpoints.latest_checkpoint(checkpoint_dir) if checkpoint_dir else None


def _frame_offsets(t_frames, t_units, t_start_obs, t_injection, device):
    """float64 tM0 per frame (emission.py:176-201 unit handling)."""
    if isinstance(t_frames, torch.Tensor) and t_frames.dtype == torch.float64 and t_frames.is_cuda \
            and getattr(t_frames, '_bhn_is_tM0', False):
        return t_frames, False
    if units.is_quantity(t_start_obs):
        t_units = t_start_obs.unit
        t_start_obs = float(t_start_obs.value)
    GM_c3 = constants.GM_c3(t_units) if t_units is not None else 1.0
    if units.is_quantity(t_frames):
        t_frames = t_frames.to(t_units).value
    if isinstance(t_frames, torch.Tensor):
        t_frames = t_frames.detach().cpu().numpy()
    tf = np.asarray(t_frames, dtype=np.float64)
    if isinstance(t_injection, torch.Tensor):
        t_injection = float(t_injection)
    return engine.frame_offsets(tf, t_start_obs, t_injection, GM_c3, device), tf.ndim == 0


def _stokes_or_none(J):
    """None for the unpolarised J=1.0 of raytracing_args (network.py:850), else J itself."""
    if J is None or (np.ndim(J) == 0 and float(J) == 1.0):
        return None
    return J


def _bound_predictor(fn):
    owner = getattr(fn, '__self__', None)
    return owner if isinstance(owner, NeRF_Predictor) else None


def _image_shape(images, B, S, sp, scalar_t):
    """(B,Sx,R) -> the reference's output shape, incl. the jnp.squeeze quirk (network.py:418)."""
    if S == 0:
        out = images.reshape((B,) + sp)
        return out.reshape(sp) if scalar_t else out
    out = images.reshape((B, S) + sp)
    return torch.squeeze(out) if not scalar_t else torch.squeeze(out.reshape((S,) + sp))


def image_plane_prediction(params, predictor_fn, t_frames, coords, Omega, J, g, dtau, Sigma, t_start_obs, t_geos,
                           t_injection, t_units):
    """Predict image pixels from emission (network.py:373-420).  With a ``NeRF_Predictor.apply``
    as ``predictor_fn`` the whole chain runs in one fused kernel and is differentiable w.r.t.
    ``params`` (the fused backward); any other callable is evaluated then integrated with the
    stand-alone radiative-transfer kernel."""
    pred = _bound_predictor(predictor_fn)
    if pred is None:
        from . import kgeo
        emission = predictor_fn({'params': params}, t_frames, t_units, coords, Omega, t_start_obs, t_geos, t_injection)
        if not np.isscalar(J):
            Jt = _hip.as_f32(J, emission.device)
            emission = utils.expand_dims(Jt, emission.ndim + 1, 0) * utils.expand_dims(emission, emission.ndim + 1, 1)
            emission = torch.squeeze(emission)
        return kgeo.radiative_trasfer(emission, g, dtau, Sigma)
    eng = pred.engine()
    geom = pred.geometry(coords, Omega, t_geos, _stokes_or_none(J), g, dtau, Sigma)
    tM0, scalar_t = _frame_offsets(t_frames, t_units, t_start_obs, t_injection, eng.device)
    flat = pred.flat_params(params)
    images = engine.RenderFunction.apply(flat, eng, geom, tM0)
    return _image_shape(images, int(tM0.numel()), geom.S, geom.spatial, scalar_t)


def loss_fn_image(params, predictor_fn, target, sigma, offset, t_frames, coords, Omega, J, g, dtau, Sigma,
                  t_start_obs, t_geos, t_injection, scale, t_units, dtype):
    """L2 loss on image pixels ('full') or light curves ('lc') (network.py:422-484)."""
    images = image_plane_prediction(params, predictor_fn, t_frames, coords, Omega, J, g, dtau, Sigma, t_start_obs,
                                    t_geos, t_injection, t_units)
    dev = images.device
    target, sigma, offset = (_hip.as_f32(v, dev) for v in (target, sigma, offset))
    if dtype == 'full':
        loss = torch.sum(torch.abs((images - target - offset) / sigma) ** 2)
    elif dtype == 'lc':
        lightcurve = images.sum(dim=(-1, -2))
        loss = torch.sum(torch.abs((lightcurve - target - offset) / sigma) ** 2)
    else:
        raise AttributeError('image dtype ({}) not supported'.format(dtype))
    return scale * loss, [images]


def loss_fn_eht(params, predictor_fn, target, sigma, A, t_frames, coords, Omega, J, g, dtau, Sigma, t_start_obs,
                t_geos, t_injection, scale, t_units, dtype):
    """chi-square loss for EHT observations: 'vis', 'amp' or 'cphase' (network.py:486-564).  ``A`` holds
    the DFT matrices per frame: (b, nvis, H*W) ('vis'/'amp') or (b, 3, nvis, H*W) ('cphase')."""
    images = image_plane_prediction(params, predictor_fn, t_frames, coords, Omega, J, g, dtau, Sigma, t_start_obs,
                                    t_geos, t_injection, t_units)
    if dtype not in engine.EHT_DTYPES:
        raise AttributeError('eht dtype ({}) not supported'.format(dtype))
    loss = engine.EhtChi2Function.apply(images.reshape(tuple(images.shape[:-2]) + (-1,)), A, target, sigma, scale, dtype)
    return loss, [images]


def dp_allreduce(buf, n, loss, rank, world):
    """The one collective of a training step (network.py:620): ``buf[:n]`` holds this rank's gradient
    of its per-device chi^2 SUM; slots ``buf[n:n+world]`` carry the per-rank losses so that a single
    all-reduce(sum) returns both.  The caller divides the gradient by ``world`` (mean over devices,
    the reference's pmean).  Returns the vector of per-rank losses (shape (world,))."""
    if world == 1 and not _dist_on():
        return loss
    import torch.distributed as dist        # (an initialised process group of ONE rank still runs the collective: the RCCL
    buf[n:].zero_()                         #  path of a single-GPU box, tests/test_gpu_ddp.py)
    buf[n + rank] = loss.reshape(-1)[0]
    dist.all_reduce(buf)                                  # RCCL over xGMI on GPUs; gloo in the CPU tests
    return buf[n:].clone()


def _exchange_and_apply(state, buf, n, loss, rank, world):
    """jax.lax.pmean(grads) + apply_gradients (network.py:620-621): one all-reduce, Adam with grad / world."""
    if (world > 1 or _dist_on()) and getattr(state, 'overlap_allreduce', False):
        return state.exchange_overlapped(buf, n, loss, rank, world)
    loss_vec = dp_allreduce(buf, n, loss, rank, world)
    state.apply_gradients(buf[:n], grad_scale=1.0 / world)
    return loss_vec


def _step_image(state, t_units, dtype, target, sigma, offset, t_frames, coords, Omega, J, g, dtau, Sigma,
                t_start_obs, t_geos, t_injection, scale, train, eht=False):
    """Per-process body of gradient_step_image/_eht and test_image/_eht with no torch.autograd in the
    loop: pack -> fused render -> chi^2 kernel -> fused backward -> all-reduce -> Adam.  For the EHT
    losses ``offset`` carries the DFT matrices A."""
    if eht:
        if dtype not in engine.EHT_DTYPES:
            raise AttributeError('eht dtype ({}) not supported'.format(dtype))
    elif dtype not in ('full', 'lc'):
        raise AttributeError('image dtype ({}) not supported'.format(dtype))
    pred = state.predictor
    eng = pred.engine()
    dev = eng.device
    geom = pred.geometry(coords, Omega, t_geos, _stokes_or_none(J), g, dtau, Sigma)
    tM0, _ = _frame_offsets(t_frames, t_units, t_start_obs, t_injection, dev)
    B = int(tM0.numel())
    eng.pack(state.flat)
    taped = train and eng.fits_tape(B, geom.P_eff)      # record the tape while rendering: no recompute later
    group = eng.tape_group(B, geom.P_eff) if (train and not taped and not eht) else 0
    if group:
        # the tape of all B frames does not fit, but chi^2 is a sum of per-frame terms: frame groups on the taped path
        tshape = (B, geom.Sx, geom.R) if dtype == 'full' else (B, geom.Sx)
        tgt, sig, off = (_hip.as_f32(v, dev).reshape(tshape) for v in (target, sigma, offset))
        images = torch.empty((B, geom.Sx, geom.R), dtype=torch.float32, device=dev)
        n = eng.nparams
        buf = state.grad_buffer()
        part = torch.empty((n,), dtype=torch.float32, device=dev)
        loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        for b0 in range(0, B, group):
            sl = slice(b0, min(b0 + group, B))
            eng.render_train(geom, tM0[sl], out=images[sl])
            lg, dimg = engine.chi2_image(images[sl], tgt[sl].contiguous(), sig[sl].contiguous(), off[sl].contiguous(), scale, dtype)
            loss += lg
            if b0 == 0:
                eng.render_bwd_tape(geom, tM0[sl], dimg, out=buf[:n])
            else:
                eng.render_bwd_tape(geom, tM0[sl], dimg, out=part)
                buf[:n] += part
        rank, world = _world()
        loss_vec = _exchange_and_apply(state, buf, n, loss, rank, world)
        out = images.reshape((1, B) + ((geom.S,) if geom.S else ()) + geom.spatial)
        return loss_vec, state, out
    images = eng.render_train(geom, tM0) if taped else eng.render(geom, tM0)
    if eht:
        loss, dimg = engine.chi2_eht(images if geom.S else images[:, 0], offset, target, sigma, scale, dtype, want_grad=train)
        dimg = dimg.reshape(images.shape) if train else None
    else:
        tshape = (B, geom.Sx, geom.R) if dtype == 'full' else (B, geom.Sx)
        tgt, sig, off = (_hip.as_f32(v, dev).reshape(tshape) for v in (target, sigma, offset))
        loss, dimg = engine.chi2_image(images, tgt, sig, off, scale, dtype, want_grad=train)
    rank, world = _world()
    if train:
        n = eng.nparams
        buf = state.grad_buffer()
        (eng.render_bwd_tape if taped else eng.render_bwd)(geom, tM0, dimg, out=buf[:n])
        loss_vec = _exchange_and_apply(state, buf, n, loss, rank, world)
    else:
        if world > 1:
            import torch.distributed as dist
            parts = [torch.empty_like(loss) for _ in range(world)]
            dist.all_gather(parts, loss)
            loss_vec = torch.cat(parts)
        else:
            loss_vec = loss
    # leading axis = this process's "device" slot of the reference's pmap output
    out = images.reshape((1, B) + ((geom.S,) if geom.S else ()) + geom.spatial)
    return loss_vec, state, out


def gradient_step_image(state, t_units, dtype, target, sigma, offset, t_frames, coords, Omega, J, g, dtau, Sigma,
                        t_start_obs, t_geos, t_injection, scale):
    """value_and_grad + pmean + apply_gradients (network.py:566-622).  Returns (loss[ndev], state,
    images[1, b_local, [S], H, W]); loss holds every rank's per-device chi^2 sum."""
    return _step_image(state, t_units, dtype, target, sigma, offset, t_frames, coords, Omega, J, g, dtau, Sigma,
                       t_start_obs, t_geos, t_injection, scale, True)


def test_image(state, t_units, dtype, target, sigma, offset, t_frames, coords, Omega, J, g, dtau, Sigma,
               t_start_obs, t_geos, t_injection, scale):
    """Forward-only twin of gradient_step_image (network.py:684-739)."""
    return _step_image(state, t_units, dtype, target, sigma, offset, t_frames, coords, Omega, J, g, dtau, Sigma,
                       t_start_obs, t_geos, t_injection, scale, False)


test_image.__test__ = False   # not a pytest test


def gradient_step_eht(state, t_units, dtype, target, sigma, A, t_frames, coords, Omega, J, g, dtau, Sigma,
                      t_start_obs, t_geos, t_injection, scale):
    """Gradient step on EHT observables (network.py:624-682); same contract as gradient_step_image."""
    return _step_image(state, t_units, dtype, target, sigma, A, t_frames, coords, Omega, J, g, dtau, Sigma,
                       t_start_obs, t_geos, t_injection, scale, True, eht=True)


def test_eht(state, t_units, dtype, target, sigma, A, t_frames, coords, Omega, J, g, dtau, Sigma,
             t_start_obs, t_geos, t_injection, scale):
    """Forward-only twin of gradient_step_eht (network.py:741-795)."""
    return _step_image(state, t_units, dtype, target, sigma, A, t_frames, coords, Omega, J, g, dtau, Sigma,
                       t_start_obs, t_geos, t_injection, scale, False, eht=True)


test_eht.__test__ = False


def sample_3d_grid(apply_fn, params, t_frame=0, t_start_obs=0, Omega=0, fov=None, coords=None, resolution=64, chunk=-1):
    """Sample the network on a 3-D grid (network.py:797-840)."""
    if (coords is None) and (fov is not None):
        grid_1d = np.linspace(-fov / 2, fov / 2, resolution)
        coords = np.array(np.meshgrid(grid_1d, grid_1d, grid_1d, indexing='ij'))
    elif coords is None:
        raise AttributeError('Either coords or fov+resolution must be provided')
    t_units = t_frame.unit if units.is_quantity(t_frame) else None
    resolution = coords.shape[-1]
    chunk = resolution if chunk < 0 else chunk
    emission = []
    for c in range(resolution // chunk):
        coords_chunk = np.ascontiguousarray(coords[:, c * chunk:(c + 1) * chunk, :, :])
        Omega_chunk = Omega if np.isscalar(Omega) else Omega[c * chunk:(c + 1) * chunk, :, :]
        emission.append(apply_fn({'params': params}, t_frame, t_units, coords_chunk, Omega_chunk, t_start_obs, 0.0, 0.0))
    return torch.cat(emission, dim=0).cpu().numpy()


def sample_checkpoint_3d(checkpoint_dir, t_frame=0, t_start_obs=0, Omega=0, fov=None, coords=None, resolution=64, chunk=-1,
                         **predictor_kw):
    """sample_3d_grid of the newest checkpoint in ``checkpoint_dir`` (network.py:842-848)."""
    predictor = NeRF_Predictor.from_yml(checkpoint_dir, **predictor_kw)
    state = checkpoints.restore_checkpoint(checkpoint_dir, None)
    if state is None:
        raise FileNotFoundError('no checkpoint in {}'.format(checkpoint_dir))
    if state.get('_legacy'):
        params = ParamTree(predictor.engine().unflatten(_hip.as_f32(state['params'], predictor.engine().device)))
    else:
        params = state['params']
    return sample_3d_grid(predictor.apply, params, t_frame, t_start_obs, Omega, fov, coords, resolution, chunk)


def image_plane_checkpoint(raytracing_args, checkpoint_dir, t, rmin=0.0, rmax=np.inf, batchsize=20, **predictor_kw):
    """Image-plane movie (nt, S, H, W) rendered from the newest checkpoint (network.py:896-906)."""
    from . import optimization
    predictor = NeRF_Predictor.from_yml(checkpoint_dir, **predictor_kw)
    predictor.rmax = min(rmax, predictor.rmax)
    predictor.rmin = max(rmin, predictor.rmin)
    params = predictor.init_params(raytracing_args)
    state = predictor.init_state(params, checkpoint_dir=checkpoint_dir)
    first = raytracing_args[0] if isinstance(raytracing_args, (list, tuple)) else raytracing_args
    num_stokes = np.shape(first['J'])[0]
    train_step = optimization.TrainStep.image(t, np.zeros((len(t), num_stokes)), dtype='lc')
    _, image_plane = optimization.total_movie_loss(batchsize, state, train_step, raytracing_args, return_frames=True)
    return image_plane


def raytracing_args(geos, Omega, t_injection, t_start_obs, J=1.0):
    """Ordered dict of the non-optimised ray-tracing arguments (network.py:850-894).

    ``geos`` is any mapping/object with x, y, z, dtau, Sigma, t.  When it is a traced geodesic record
    (``kgeo.image_plane_geos``: r, theta, affine, potentials, constants of motion) the Doppler factor is derived from
    the azimuthal 4-velocity of ``Omega`` as the reference does (network.py:875-876); a pre-computed table may carry
    its own ``g`` instead."""
    from . import kgeo
    get = (lambda k: geos[k]) if isinstance(geos, dict) else (lambda k: getattr(geos, k))

    def has(k):
        try:
            get(k)
            return True
        except (KeyError, AttributeError):
            return False

    if all(has(k) for k in ('r', 'theta', 'affine', 'R', 'Theta', 'Delta', 'Xi', 'lam', 'E', 'M', 'spin')):
        gfac = kgeo.doppler_factor(geos, kgeo.azimuthal_velocity_vector(geos, Omega))
    elif has('g'):
        gfac = get('g')
    else:
        raise AttributeError('geos must be a traced geodesic record (kgeo.image_plane_geos) or carry the Doppler factor "g"')
    f32 = lambda v: np.ascontiguousarray(np.asarray(v, dtype=np.float32))
    return OrderedDict({
        'coords': f32(np.array([get('x'), get('y'), get('z')])),
        'Omega': f32(Omega),
        'J': J if np.isscalar(J) else f32(J),
        'g': f32(gfac),
        'dtau': f32(get('dtau')),
        'Sigma': f32(get('Sigma')),
        't_start_obs': t_start_obs,
        't_geos': f32(get('t')),
        't_injection': t_injection})
