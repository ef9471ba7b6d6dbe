"""bhnerf.optimization on one-process-per-GPU (reference: bhnerf/optimization.py).

The reference drives ``jax.pmap`` from a single process: a batch of frame indices is reshaped to
(ndev, b/ndev, ...) (optimization.py:289-291, 360-362) and gradients are ``pmean``-ed
(network.py:620).  Here every GPU has its own process (``torch.distributed``, backend "nccl" = RCCL
over xGMI): rank r takes the r-th contiguous slice of the same batch, the flat gradient buffer is
all-reduced once per step and divided by the world size, and every rank applies the identical Adam
step -- the same mean-of-per-device-sums arithmetic.  All ranks must draw the same batch indices:
``TemporalBatchedArgs.sample`` uses a generator seeded identically on every rank.
"""
import os

import numpy as np
import torch

from . import checkpoints, network, units

try:  # progress bar is optional
    from tqdm.auto import tqdm
except Exception:  # pragma: no cover
    tqdm = lambda x, **kw: x


class HostReadable(torch.Tensor):
    """Device tensor that NumPy can read.  The reference's drivers log ``np.log10(np.mean(opt.loss))`` on a JAX device
    array (scripts/Fit_ALMA_LP_Apr11_SgrA_Flare.py); ``Optimizer.loss`` stays on the GPU (no per-step sync) and is copied
    to the host only when NumPy asks for it."""

    def __array__(self, dtype=None, copy=None):
        host = self.detach().as_subclass(torch.Tensor).cpu().numpy()
        return host.astype(dtype) if dtype is not None else host

    def __array_wrap__(self, array, context=None, return_scalar=False):     # results of NumPy ufuncs stay NumPy
        return array[()] if return_scalar or np.ndim(array) == 0 else array

    def mean(self, *args, axis=None, out=None, **kw):       # np.mean(x) calls x.mean(axis=None, dtype=None, out=None)
        if axis is not None:
            kw['dim'] = axis
        if kw.get('dtype', 0) is None:
            kw.pop('dtype')
        return super().mean(*args, **kw)


def device_count():
    """Number of data-parallel workers (the reference's jax.device_count())."""
    return network._world()[1]


def shard(xs):
    """This rank's contiguous slice of the leading (batch) axis (optimization.py:360-362)."""
    rank, world = network._world()
    def one(x):
        if x.shape[0] % world:
            raise ValueError('batch size {} is not divisible by the number of devices {}'.format(x.shape[0], world))
        n = x.shape[0] // world
        return x[rank * n:(rank + 1) * n]
    if isinstance(xs, (list, tuple)):
        return [one(x) for x in xs]
    return one(xs)


def _movie_chunks(nt, batchsize, ndev):
    """Frame-index chunks of a whole-movie pass: full batches first, then one remainder chunk padded
    (wrapping around the movie) to a multiple of the device count -- the split of optimization.py:42-46."""
    if nt % ndev:
        raise AttributeError('batch size should be an integer multiplication of the device number')
    full = (nt // batchsize) * batchsize
    chunks = [np.arange(lo, lo + batchsize) for lo in range(0, full, batchsize)]
    padded_end = ndev * int(np.ceil(nt / ndev))
    tail = np.arange(full, padded_end) % nt
    return chunks + ([tail] if tail.size else [])


def total_movie_loss(batchsize, state, train_step, raytracing_args, return_frames=False):
    """Loss of the whole movie in test mode, evaluated in frame batches that fit the GPUs; optionally also the
    predicted frames (optimization.py:14-66).  Returns total_loss / nt [, frames (nt,[S],H,W) NumPy]."""
    nt = train_step.args[0].num_frames
    ndev = device_count()
    polarised = not np.isscalar(np.atleast_1d(raytracing_args)[0]['J'])
    total, movie = 0.0, []
    for inds in _movie_chunks(nt, batchsize, ndev):
        loss, state, images = train_step(state, raytracing_args, inds, update_state=False)
        total = total + loss.sum()                            # stays on the device: ONE host sync per movie, not per batch
        if not return_frames:
            continue
        if ndev > 1:                                          # every rank rendered its slice of the chunk
            import torch.distributed as dist
            parts = [torch.empty_like(images) for _ in range(ndev)]
            dist.all_gather(parts, images.contiguous())
            images = torch.cat(parts, dim=0)
        keep = 3 if polarised else 2                          # trailing ([S],H,W) axes of one frame
        movie.append(images.reshape((-1,) + tuple(images.shape[-keep:])).cpu().numpy())
    mean_loss = float(total) / nt
    return (mean_loss, np.concatenate(movie)[:nt]) if return_frames else mean_loss


class Optimizer(object):
    """Adam optimisation of the network parameters over random frame batches (optimization.py:68-143).

    hparams: 'num_iters'; optional 'lr_init' (1e-4), 'lr_final' (1e-6), 'lr_inject' (None, inert as in the
    reference), 'seed' (1).  save_period < 0 saves only the final step; keep = checkpoints retained."""

    def __init__(self, hparams, predictor, raytracing_args, save_period=-1, checkpoint_dir='', keep=5):
        self.num_iters = hparams['num_iters']
        self.seed = hparams.get('seed', 1)
        self.checkpoint_dir, self.keep = checkpoint_dir, keep
        self.save_period = save_period if save_period >= 0 else self.num_iters
        self.step = self.init_step = 0
        self.loss = np.inf
        self.log_fns = []
        self.state = predictor.init_state(
            params=predictor.init_params(raytracing_args, seed=self.seed), num_iters=self.num_iters,
            lr_init=hparams.get('lr_init', 1e-4), lr_final=hparams.get('lr_final', 1e-6),
            lr_inject=hparams.get('lr_inject', None), checkpoint_dir=checkpoint_dir)
        # opt-in: all-reduce of step k under the forward / backward of step k+1 (one-step-stale gradients; network.TrainState)
        self.state.overlap_allreduce = bool(hparams.get('overlap_allreduce', False))
        self.hip_graph = hparams.get('hip_graph', None)       # None: leave the TrainStep's own setting (BHNERF_HIP_GRAPH)
        if checkpoint_dir and network._world()[0] == 0:
            predictor.save_params(checkpoint_dir)

    def log(self):
        for fn in self.log_fns:
            fn(self)

    def save_checkpoint(self):
        due = self.step % self.save_period == 0 or self.step == self.final_step
        if self.checkpoint_dir and due:
            self.state.finish_allreduce()       # (overlap_allreduce: every rank applies the gradient in flight, then rank 0 writes)
        if not (self.checkpoint_dir and due and network._world()[0] == 0):
            return
        # the flax file format of the reference (optimization.py:118-121), see checkpoints.py
        checkpoints.save_checkpoint(self.checkpoint_dir, self.state, int(self.step), keep=self.keep, overwrite=True)

    def run(self, batchsize, train_step, raytracing_args, log_fns=[]):
        """num_iters more steps from wherever the (possibly restored) state stands; Ctrl-C stops early."""
        self.init_step = int(self.state.step) + 1
        self.final_step = self.init_step + self.num_iters
        self.log_fns = list(np.atleast_1d(log_fns))
        self.train_step, self.raytracing_args = train_step, raytracing_args
        if self.hip_graph is not None:
            train_step.use_graph = bool(self.hip_graph)
        frames = train_step.args[0]
        bar = tqdm(range(self.init_step, self.final_step), desc='iteration', disable=network._world()[0] != 0)
        try:
            for self.step in bar:
                loss, self.state, _ = train_step(self.state, raytracing_args, indices=frames.sample(batchsize))
                self.loss = loss.as_subclass(HostReadable) if isinstance(loss, torch.Tensor) else loss
                self.log()
                self.save_checkpoint()
        except KeyboardInterrupt:
            pass
        finally:
            # overlap_allreduce: the gradient still in flight is completed and applied on every exit path (Ctrl-C
            # included), so that opt.params / a later save never lag one update behind
            self.state.finish_allreduce()

    @property
    def params(self):
        return self.state.params


def _shared_rng(stream):
    """Generator that is identical on every rank (BHNERF_BATCH_SEED, default 0; `stream` separates its users): all
    ranks must draw the same frame batches and the same ray set, as the reference's one process does."""
    return np.random.default_rng([int(os.environ.get('BHNERF_BATCH_SEED', '0')), int(stream)])


class GraphedImageStep:
    """One `gradient_step_image` (pack -> training forward -> chi^2 -> fused backward -> [all-reduce] -> Adam) for a fixed ray
    set and batch size, captured ONCE into a HIP graph and replayed per step.  Everything that changes from step to step lives
    in device buffers the graph reads: the frame indices (the batch of target / sigma / offset is gathered inside the graph),
    the frame time offsets tM0, and Adam's learning rate and bias corrections (bhn_adam_step_dev).  Per step the host fills one
    pinned staging buffer, issues ONE small asynchronous copy and one graph launch -- instead of ~25 Python-level launches --
    which is what a GPU needs when its share of a step is a single small frame (the reference's b = 8 on 8 devices,
    optimization.py:289-291).  With an RCCL process group the all-reduce and Adam are captured too (`collective_in_graph`);
    when the backend cannot be captured (gloo) the graph ends at the gradient and the exchange + Adam follow eagerly.
    The arithmetic is the un-captured step's, kernel for kernel: parameters are bitwise equal (tests/test_gpu_api.py).

    A HIP graph bakes raw device addresses in.  The step therefore keeps a reference to everything it captured (ray-tracing
    dict, geometry, workspace, parameters, Adam moments, gradient buffer) and `stale()` compares them with what an eager step
    would use NOW -- a re-allocated workspace (another ray set or batch size needed a larger one), a replaced parameter or
    gradient tensor, an edited ray set (new geometry fingerprint), a cleared geometry cache -- before every replay; a stale
    graph is dropped and captured again.  `loss` is returned as a copy; `images` is a VIEW of the graph's output buffer, which
    the next step on this ray set overwrites (the reference's training loop discards it: optimization.py:132)."""

    def __init__(self, state, args, dtype, scale, rt, n_local):
        from . import engine, _hip
        pred = state.predictor
        eng = pred.engine()
        self.state, self.eng, self.args, self.dtype, self.scale = state, eng, args, dtype, float(scale)
        self.rt, self.pred = rt, pred                                          # (kept alive: the cache key holds their ids)
        self.epoch = getattr(pred, '_graph_epoch', 0)
        dev = eng.device
        self.geom = geom = self._geometry()
        B = self.B = int(n_local)
        if not eng.fits_tape(B, geom.P_eff):
            raise ValueError('the tape of %d frames does not fit the workspace: no graph for this step' % B)
        t_start, t_units = rt['t_start_obs'], args.t_units
        if units.is_quantity(t_start):
            t_units, t_start = t_start.unit, float(t_start.value)
        from . import constants
        self.GM = constants.GM_c3(t_units) if t_units is not None else 1.0
        self.t_start, self.t_inj = float(t_start), float(rt['t_injection'])
        args[np.arange(min(args.num_frames, network._world()[1]))]            # (makes the device-resident copies of the per-frame arrays)
        # whole-movie target, sigma, offset as ONE tensor (the batch is gathered by a single index_select): the stack
        # TemporalBatchedArgs keeps for its own gathers -- no third device copy of the movie
        if args._stack is None or args._stack.shape[0] != 3 or args._stack.dtype != torch.float32:
            raise ValueError('target / sigma / offset of one shape and dtype expected: no graph for this step')
        self.full = args._stack
        self.tshape = (B, geom.Sx, geom.R) if dtype == 'full' else (B, geom.Sx)
        # per-step inputs: a ring of pinned staging slots (the host runs ahead of the GPU: a slot is rewritten only after the
        # copies issued from it have executed), three device buffers the graph reads
        # (one buffer [idx: B int64 | tM0: B float64 | lr, 1-b1^t, 1-b2^t, pad: 4 float32] = ONE copy per step; the host writes
        #  it through NumPy views: every torch call here costs the driver microseconds it does not have at one frame per GPU)
        nbytes = 16 * B + 16
        self.slots = []
        for _ in range(8):
            h = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
            a = h.numpy()
            self.slots.append(dict(buf=h, idx=a[:8 * B].view(np.int64), tM0=a[8 * B:16 * B].view(np.float64),
                                   hyp=a[16 * B:16 * B + 12].view(np.float32), done=torch.cuda.Event()))
        self.slot_i, self.slot_used = 0, 0
        self.d_buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.d_idx = self.d_buf[:8 * B].view(torch.int64)
        self.d_tM0 = self.d_buf[8 * B:16 * B].view(torch.float64)
        self.d_hyp = self.d_buf[16 * B:16 * B + 16].view(torch.float32)
        self.d_hyp.fill_(1.0)
        self.images = torch.empty((B, geom.Sx, geom.R), dtype=torch.float32, device=dev)
        self.n = eng.nparams
        # a process group: the all-reduce is captured when the backend runs on the stream (RCCL); else exchange + Adam stay eager
        self.dist = network._dist_on()
        self.collective_in_graph = self.dist and self._backend() == 'nccl' and os.environ.get('BHNERF_GRAPH_COLLECTIVE', '1') == '1'
        self.with_adam = (not self.dist) or self.collective_in_graph
        self.graph = None
        self.loss = self.lossv = None
        self.captured = None

    @staticmethod
    def _backend():
        import torch.distributed as dist
        try:
            return dist.get_backend()
        except Exception:
            return None

    def _geometry(self):
        rt = self.rt
        return self.pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], network._stokes_or_none(rt['J']), rt['g'], rt['dtau'], rt['Sigma'])

    def matches(self, state, rt):
        """This entry was built for exactly these objects (the cache key is their id(): ids are only unique among LIVE
        objects, and this entry keeps its own alive)."""
        return state is self.state and rt is self.rt

    def stale(self):
        """True when an eager step would no longer use what this graph captured (see the class docstring)."""
        st, eng = self.state, self.eng
        if self.pred.engine() is not eng or getattr(self.pred, '_graph_epoch', 0) != self.epoch:
            return True
        if self._geometry() is not self.geom:                  # edited ray set / evicted and rebuilt geometry
            return True
        rt = self.rt
        t_start = rt['t_start_obs']
        if float(getattr(t_start, 'value', t_start)) != self.t_start or float(rt['t_injection']) != self.t_inj:
            return True
        if self.captured is None:
            return False
        ws, flat, m, v, grad = self.captured
        return not (eng._ws is ws and st.flat is flat and st.m is m and st.v is v and st.grad_buffer() is grad)

    def _body(self):
        from . import engine
        st, eng, geom = self.state, self.eng, self.geom
        eng.pack(st.flat)
        tgt, sig, off = (a.reshape(self.tshape) for a in self.full.index_select(1, self.d_idx))
        images = eng.render_train(geom, self.d_tM0, out=self.images)
        loss, dimg = engine.chi2_image(images, tgt, sig, off, self.scale, self.dtype)
        buf = st.grad_buffer()
        eng.render_bwd_tape(geom, self.d_tM0, dimg, out=buf[:self.n])
        lossv = loss
        if self.collective_in_graph:
            rank, world = network._world()
            lossv = network.dp_allreduce(buf, self.n, loss, rank, world)            # (the reference's pmean, network.py:620)
            engine.adam_step_dev(st.flat, buf[:self.n], st.m, st.v, self.d_hyp, grad_scale=1.0 / world)
        elif self.with_adam:
            engine.adam_step_dev(st.flat, buf[:self.n], st.m, st.v, self.d_hyp, grad_scale=1.0)
        return loss, lossv, buf

    def _stage(self, key):
        from . import engine
        st = self.state
        sl = self.slots[self.slot_i]
        self.slot_i = (self.slot_i + 1) % len(self.slots)
        if self.slot_used >= len(self.slots):
            sl['done'].synchronize()                       # (waits only when the GPU is a whole ring behind the host)
        self.slot_used += 1
        sl['idx'][:] = key
        sl['tM0'][:] = (self.args.t_values[key] - self.t_start) / self.GM - self.t_inj          # engine.frame_offsets, float64
        sl['hyp'][:] = engine.adam_hyper(st.step + 1, st.learning_rate())
        self.d_buf.copy_(sl['buf'], non_blocking=True)
        sl['done'].record()

    def _capture(self):
        st = self.state
        # two eager steps' worth of kernels on a side stream first (lazy allocations, kernel attributes, the communicator's
        # first collective), on a COPY of the optimiser state; then the capture
        keep = [t.clone() for t in (st.flat, st.m, st.v)]
        s = torch.cuda.Stream(device=self.eng.device)
        s.wait_stream(torch.cuda.current_stream(self.eng.device))
        with torch.cuda.stream(s):
            self._body()
        torch.cuda.current_stream(self.eng.device).wait_stream(s)
        for t, k in zip((st.flat, st.m, st.v), keep):
            t.copy_(k)
        graph = torch.cuda.CUDAGraph()
        captured, why = True, None
        try:
            with torch.cuda.graph(graph):
                self.loss, self.lossv, self.buf = self._body()
        except RuntimeError as exc:                       # (what a failed stream capture raises; anything else is a bug and propagates)
            if not self.collective_in_graph:
                raise
            captured, why = False, exc
        if self.collective_in_graph:
            # The ranks must AGREE on where the exchange runs: a rank that fell back on its own would issue the collective
            # eagerly while the others replay it inside their graphs -- another operation order, a hang.  One eager all-reduce
            # (MIN) of the outcome; if any rank could not capture the collective, every rank keeps the exchange and Adam
            # outside the graph and captures again.
            rank, world = network._world()
            if world > 1:
                import torch.distributed as dist
                flag = torch.tensor([1.0 if captured else 0.0], device=self.eng.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                everywhere = bool(flag.item() > 0)
            else:
                everywhere = captured
            if not everywhere:
                import warnings
                warnings.warn('hip_graph: the gradient all-reduce could not be captured into the step graph on %s (%r); the exchange and '
                              'Adam stay outside the graph on every rank' % ('this rank' if not captured else 'another rank', why))
                self.collective_in_graph, self.with_adam = False, False
                torch.cuda.synchronize(self.eng.device)
                for t, k in zip((st.flat, st.m, st.v), keep):
                    t.copy_(k)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.loss, self.lossv, self.buf = self._body()
        for t, k in zip((st.flat, st.m, st.v), keep):                          # (capture does not execute, but stay safe)
            t.copy_(k)
        self.graph = graph
        self.captured = (self.eng._ws, st.flat, st.m, st.v, self.buf)

    def __call__(self, key):
        """key: this rank's frame indices of the step (length B).  Returns (loss vector, state, images (1, B, [S], H, W))."""
        st, geom = self.state, self.geom
        assert len(key) == self.B
        self._stage(key)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        rank, world = network._world()
        if self.with_adam:
            st.step += 1
            loss_vec = self.lossv.clone()
        else:
            loss_vec = network._exchange_and_apply(st, self.buf, self.n, self.loss, rank, world)
        out = self.images.reshape((1, self.B) + ((geom.S,) if geom.S else ()) + geom.spatial)
        return loss_vec, st, out


class TrainStep(object):
    """Container of per-loss step functions (optimization.py:145-272)."""

    def __init__(self, dtype, args, grad_pmap, test_pmap, scale):
        self.dtype = np.atleast_1d(dtype)
        self.args = np.atleast_1d(args)
        self.grad_pmap = np.atleast_1d(grad_pmap)
        self.test_pmap = np.atleast_1d(test_pmap)
        self.scale = np.atleast_1d(scale)
        if np.any([units.unit_name(arg.t_units) not in ('hr', 'h') for arg in self.args]):
            raise AttributeError('only hr units supported')
        assert self.dtype.size == self.args.size == self.test_pmap.size == \
            self.grad_pmap.size == self.scale.size, 'input list sizes are not equal'
        self.num_losses = self.dtype.size
        # the sub-pixel ray set of a training step is ONE choice for all devices (optimization.py:169 runs once in the
        # reference's single process): every rank draws it from a generator seeded like TemporalBatchedArgs._rng
        self._rng = _shared_rng(1)
        # opt-in (hparams['hip_graph'] / BHNERF_HIP_GRAPH=1): training steps of a single image-plane loss are captured into
        # one HIP graph per (ray set, batch size) -- GraphedImageStep
        self.use_graph = os.environ.get('BHNERF_HIP_GRAPH') == '1'
        self._graphs = {}

    MAX_GRAPHS = 16          # > the sub-pixel ray sets of a run (scripts use up to 10); each entry owns a HIP graph + staging

    def clear_graphs(self):
        """Drop every captured step (they are rebuilt on demand)."""
        self._graphs.clear()

    def _graphed(self, state, rt, indices):
        key = shard(np.atleast_1d(np.asarray(indices)))
        gkey = (id(rt), len(key), id(state))
        g = self._graphs.get(gkey)
        if g is not None and g is not False and (not g.matches(state, rt) or g.stale()):
            # another object at a recycled id, or the step would no longer run on what was captured (re-allocated workspace,
            # replaced state tensors, edited ray set, cleared geometry cache): never replay it
            del self._graphs[gkey]
            g = None
        if g is None:
            try:
                g = GraphedImageStep(state, self.args[0], str(self.dtype[0]), float(self.scale[0]), rt, len(key))
            except ValueError:
                g = False                                  # (tape does not fit / arguments of mixed shapes: this step stays eager)
            self._graphs[gkey] = g if g else False
            if g is False:
                self._nograph_refs = getattr(self, '_nograph_refs', {})
                self._nograph_refs[gkey] = (rt, state)     # (keeps the ids of a "no graph" entry from being recycled)
            while len(self._graphs) > self.MAX_GRAPHS:
                old = next(iter(self._graphs))
                self._graphs.pop(old)
                getattr(self, '_nograph_refs', {}).pop(old, None)
        else:
            self._graphs[gkey] = self._graphs.pop(gkey)    # LRU order
        return g(key) if g else None

    def __call__(self, state, raytracing_args, indices, update_state=True):
        """One pass over the losses.  Training picks ONE ray set at random (stochastic sub-pixel sampling,
        optimization.py:169) and lets every loss take its own Adam step; testing averages over all ray sets."""
        ray_sets = list(np.atleast_1d(raytracing_args))
        if update_state:
            ray_sets = [ray_sets[int(self._rng.integers(len(ray_sets)))]]
        fns = self.grad_pmap if update_state else self.test_pmap
        if (update_state and self.use_graph and self.num_losses == 1 and fns[0] is network.gradient_step_image
                and torch.cuda.is_available() and not getattr(state, 'overlap_allreduce', False)
                and type(state.predictor) is network.NeRF_Predictor):
            res = self._graphed(state, ray_sets[0], indices)
            if res is not None:
                return res
        loss_acc = images_acc = 0.0
        for rt in ray_sets:
            for k in range(self.num_losses):
                loss, state, images = fns[k](state, self.t_units, self.dtype[k], *self.args[k][indices], *rt.values(),
                                             float(self.scale[k]))
                if len(ray_sets) == 1 and self.num_losses == 1:
                    return loss, state, images          # (x / 1 + 0.0 is x: four element-wise launches per step that compute nothing)
                loss_acc = loss_acc + loss / len(ray_sets)
                images_acc = images_acc + images / len(ray_sets)
        return loss_acc, state, images_acc

    def __add__(self, other):
        return TrainStep(np.append(self.dtype, other.dtype), np.append(self.args, other.args),
                         np.append(self.grad_pmap, other.grad_pmap), np.append(self.test_pmap, other.test_pmap),
                         np.append(self.scale, other.scale))

    @classmethod
    def image(cls, t_frames, target, sigma=1.0, offset=0.0, scale=1.0, dtype='full'):
        """Training step for image-plane measurements: 'full' pixels or 'lc' light curves."""
        target = np.asarray(target)
        sigma = sigma * np.ones_like(target)
        offset = offset * np.ones_like(target)
        args = TemporalBatchedArgs(t_frames, [target, sigma, offset])
        return cls(dtype, args, network.gradient_step_image, network.test_image, scale)

    @classmethod
    def eht_arrays(cls, t_frames, target, sigma, A, dtype='vis', scale=1.0):
        """Training step on EHT observables from plain arrays: per frame the measurement `target`
        (nt, nvis) (complex for 'vis'), its `sigma`, and the DFT matrices `A` (nt, nvis, H*W) or, for
        'cphase', (nt, 3, nvis, H*W) (radians) -- what ehtim's chisqdata_<dtype> returns per frame
        (optimization.py:237-257).  See ``observation.dft_matrix`` for a direct-DFT `A`."""
        if dtype not in ('vis', 'amp', 'cphase'):
            raise AttributeError('eht dtype ({}) not supported'.format(dtype))
        args = TemporalBatchedArgs(t_frames, [np.asarray(target), np.asarray(sigma), np.asarray(A)])
        return cls(dtype, args, network.gradient_step_eht, network.test_eht, scale)

    @classmethod
    def eht(cls, t_frames, obs, image_fov, image_size, chisqdata, pol='I', scale=1.0):
        """Training step for an ehtim observation (optimization.py:219-268); needs the external ehtim
        package to split the observation per frame and build (target, sigma, A)."""
        try:
            from ehtim.image import make_square
        except ImportError as exc:
            raise ImportError('TrainStep.eht needs ehtim; use TrainStep.eht_arrays with your own (target, sigma, A)') from exc
        dtype = chisqdata.__name__.split('_')[-1]
        for p in np.atleast_1d(pol):
            if p not in ('I', 'Q', 'U'):
                raise AttributeError('pol ({}) not in supported pol_types: I,Q,U'.format(p))
        span = (t_frames[-1] - t_frames[0]).to('s').value
        frames = obs.split_obs(t_gather=span / (len(t_frames) + 1))
        prior = make_square(obs, image_size, image_fov)
        per_pol = [[np.array(x) for x in zip(*[chisqdata(f, prior, mask=[], pol=p) for f in frames])] for p in np.atleast_1d(pol)]
        target, sigma, A = (np.squeeze(np.stack([pp[i] for pp in per_pol], axis=1)) for i in range(3))
        if dtype == 'cphase':                                   # ehtim closure phases are in degrees
            target, sigma = np.deg2rad(target), np.deg2rad(sigma)
        return cls.eht_arrays(t_frames, target, sigma, A, dtype, scale)

    @property
    def t_units(self):
        return self.args[0].t_units


class TemporalBatchedArgs(object):
    """Per-frame arrays + frame times, batched by frame index (optimization.py:274-302).  The
    arrays stay resident on the device; a batch is a device gather of this rank's index slice."""

    def __init__(self, t_frames, args=[]):
        self.t_frames = t_frames
        if not isinstance(args, list):
            args = [args]
        self.num_frames = len(t_frames)
        assert all([self.num_frames == arg.shape[0] for arg in args])
        self.host_args = [np.asarray(a, dtype=np.complex64 if np.iscomplexobj(a) else np.float32) for a in args]
        self.t_values = np.asarray(units.strip(t_frames), dtype=np.float64)
        self.args = self.host_args + [self.t_values]
        self.default_t_units = units.hr
        self._dev = None
        self._stack = None
        self._rng = _shared_rng(0)

    def sample(self, batchsize, replace=False):
        """Random frame batch; identical on every rank (same seed, same call sequence)."""
        return self._rng.choice(self.num_frames, batchsize, replace=replace)

    def __getitem__(self, key):
        key = shard(np.atleast_1d(np.asarray(key)))
        if torch.cuda.is_available():
            if self._dev is None:
                dev = torch.device('cuda', torch.cuda.current_device())
                self._dev = [torch.as_tensor(a, device=dev) for a in self.host_args]
                # arguments of one shape and dtype (target / sigma / offset of an image-plane loss) are gathered by ONE kernel
                same = len(self._dev) > 1 and len({(tuple(a.shape), a.dtype) for a in self._dev}) == 1
                self._stack = torch.stack(self._dev) if same else None
            from . import _hip
            idx = _hip.h2d_small(np.asarray(key, dtype=np.int64), self._dev[0].device if self._dev else 'cuda')
            if self._stack is not None:
                out = list(self._stack.index_select(1, idx))
            else:
                out = [a.index_select(0, idx) for a in self._dev]
        else:
            out = [a[key, ...] for a in self.host_args]
        out.append(self.t_values[key])
        return out

    @property
    def t_units(self):
        return self.t_frames.unit if units.is_quantity(self.t_frames) else self.default_t_units

    @property
    def t_start_obs(self):
        return self.t_frames[0]


class LogFn(object):
    """Call ``log_fn(optimizer)`` every ``log_period`` iterations (optimization.py:349-357)."""

    def __init__(self, log_fn, log_period=1):
        self.log_period = log_period
        self.log_fn = log_fn

    def __call__(self, optimizer):
        if self.log_period > 0:
            if (optimizer.step == 1) or ((optimizer.step % self.log_period) == 0):
                self.log_fn(optimizer)


class SummaryWriter(object):
    """Training-log writer with the methods the reference's drivers use (optimization.py:304-347, a subclass of
    ``tensorboardX.SummaryWriter`` there).  tensorboardX is used when it is installed; otherwise scalars go to
    ``<logdir>/scalars.jsonl`` (one ``{"tag", "value", "step"}`` object per line), image batches to
    ``<logdir>/<tag>_<step>.npy`` and figures to ``<logdir>/<tag>_<step>.png`` -- plain files, no event protocol."""

    def __init__(self, logdir=None, comment='', **kwargs):
        self.logdir = logdir or os.path.join('runs', comment or 'bhnerf')
        self._tb = None
        try:
            import tensorboardX
            self._tb = tensorboardX.SummaryWriter(self.logdir, comment, **kwargs)
        except ImportError:
            os.makedirs(self.logdir, exist_ok=True)

    def _path(self, tag, step, ext):
        path = os.path.join(self.logdir, '%s_%s.%s' % (tag.replace('/', '_'), step, ext))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        return path

    def add_scalar(self, tag, scalar_value, global_step=None):
        if self._tb is not None:
            return self._tb.add_scalar(tag, scalar_value, global_step)
        import json
        with open(os.path.join(self.logdir, 'scalars.jsonl'), 'a') as f:
            f.write(json.dumps({'tag': tag, 'value': float(scalar_value), 'step': None if global_step is None else int(global_step)}) + '\n')

    def add_images(self, tag, img_tensor, global_step=None, dataformats='NCHW'):
        if self._tb is not None:
            return self._tb.add_images(tag, img_tensor, global_step, dataformats=dataformats)
        np.save(self._path(tag, global_step, 'npy'), np.asarray(img_tensor))

    def add_figure(self, tag, figure, global_step=None, close=True):
        if self._tb is not None:
            return self._tb.add_figure(tag, figure, global_step, close)
        figure.savefig(self._path(tag, global_step, 'png'))
        if close:
            import matplotlib.pyplot as plt
            plt.close(figure)

    def flush(self):
        if self._tb is not None:
            self._tb.flush()

    def close(self):
        if self._tb is not None:
            self._tb.close()

    def recovery_3d(self, fov, vis_res=64, emission_true=None):
        """Log function for ``Optimizer.run(log_fns=...)``: samples the network on a ``vis_res``^3 grid of width ``fov``
        (or on the grid of ``emission_true``: an array with 1-D ``x, y, z`` coordinate attributes) and logs its slices,
        plus mse / psnr against the truth (optimization.py:310-329)."""
        from . import utils
        if emission_true is not None:
            axes = [np.linspace(float(c[0]), float(c[-1]), n) for c, n in
                    zip((emission_true.x, emission_true.y, emission_true.z), emission_true.shape[:3])]
            truth = np.asarray(getattr(emission_true, 'data', emission_true))
        else:
            axes = [np.linspace(-fov / 2.0, fov / 2.0, vis_res)] * 3
        vis_coords = np.array(np.meshgrid(*axes, indexing='ij'))

        def log_fn(opt):
            grid = network.sample_3d_grid(opt.state.apply_fn, opt.params, coords=vis_coords)
            self.add_images('emission/estimate', utils.intensity_to_nchw(grid), global_step=opt.step, dataformats='NCWH')
            if emission_true is not None:
                self.add_scalar('emission/mse', utils.mse(truth, grid), global_step=opt.step)
                self.add_scalar('emission/psnr', utils.psnr(truth, grid), global_step=opt.step)
        return log_fn

    def plot_lc_datafit(self, opt, name, train_step, target, stokes, t_frames=None, batchsize=20):
        """Figure of the estimated against the target Stokes light curves, and the log10 data-fit scalar
        (optimization.py:331-347).  One of the sub-pixel ray sets is drawn at random, as in the reference."""
        import matplotlib
        matplotlib.use('Agg', force=False)
        import matplotlib.pyplot as plt
        rt = opt.raytracing_args
        if isinstance(rt, (list, tuple)):                       # same choice on every rank (all ranks render the movie)
            rt = rt[int(_shared_rng(2 + int(opt.step)).integers(len(rt)))]
        loss, movie = total_movie_loss(batchsize, opt.state, train_step, rt, return_frames=True)
        lc_est = np.asarray(movie).sum(axis=(-1, -2))
        target = np.asarray(target)
        t = np.arange(len(target)) if t_frames is None else np.asarray(getattr(t_frames, 'value', t_frames))
        fig, axes = plt.subplots(1, len(stokes), figsize=(4 * len(stokes), 3), squeeze=False)
        for k, (ax, s) in enumerate(zip(axes[0], stokes)):
            ax.plot(t, target[:, k] if target.ndim > 1 else target, label='True')
            ax.plot(t, lc_est[:, k] if lc_est.ndim > 1 else lc_est, 'rx', label='Estimate')
            ax.set_title(s)
            ax.legend()
        self.add_figure('lightcurve/{}'.format(name), fig, global_step=opt.step)
        self.add_scalar('datafit/{}'.format(name), np.log10(np.mean(loss)), global_step=opt.step)
