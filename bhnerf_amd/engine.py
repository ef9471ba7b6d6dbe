"""Device engine: prepared ray geometry + fused predictor/render calls through the C ABI.

This is the layer the reference-shaped API (network.py / optimization.py of this package) is
built on.  Everything here runs on the HIP library; nothing falls back to the CPU.
"""
import ctypes as C
import functools

import numpy as np
import torch

from . import _hip
from .constants import GM_c3_hr

COMPACT_POINTS = True          # point-level compaction of the domain mask for the fused kernels (RayGeometry.compact)
COMPACT_BELOW = 0.9            # ... when less than this fraction of the samples is inside the domain


def _flat(t, n):
    return t.reshape(n).contiguous()


def _on_device(fn):
    """Run a method with its object's device current: the C ABI launches on the CURRENT device of the calling thread
    (kernel attributes and occupancy are cached per device), while streams and pointers come from ``self.device`` --
    a predictor built for cuda:1 must work when the process's current device is cuda:0."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *args, **kw)
    return wrapper


class RayGeometry:
    """Static per-(ray, sample) inputs resident in HBM, laid out as flat [ray][sample] planes.

    Replaces the per-iteration broadcasts of the reference (kgeo.py:618-621 g^2*dtau*Sigma,
    network.py:416-417 J, emission.py:370-373 domain mask) with a one-off fold
    (``bhn_geom_prepare``).  coords (3,*sp,G); Omega/t_geos/g/dtau/Sigma (*sp,G) or scalars;
    J None/scalar 1.0 or (S,*sp,G).
    """

    def __init__(self, coords, Omega, g, dtau, Sigma, t_geos, J=None, rmin=0.0, rmax=np.inf, z_width=np.inf,
                 device='cuda'):
        with torch.cuda.device(torch.device(device)):
            self._build(coords, Omega, g, dtau, Sigma, t_geos, J, rmin, rmax, z_width, torch.device(device))

    def _build(self, coords, Omega, g, dtau, Sigma, t_geos, J, rmin, rmax, z_width, dev):
        coords = _hip.as_f32(coords, dev)
        self.spatial = tuple(coords.shape[1:-1])
        self.G = int(coords.shape[-1])
        self.R = int(np.prod(self.spatial)) if self.spatial else 1
        self.P = self.R * self.G
        P = self.P
        full = tuple(coords.shape[1:])

        def plane(v):
            v = _hip.as_f32(v, dev)
            return _flat(v.expand(full) if v.shape != full else v, P)

        self.coords = coords.reshape(3, P).contiguous()
        self.Omega, self.t_geo = plane(Omega), plane(t_geos)
        g, dtau, Sigma = plane(g), plane(dtau), plane(Sigma)
        if J is None or np.isscalar(J) or (hasattr(J, 'ndim') and J.ndim == 0):
            if J is not None and float(J) != 1.0:
                g = g * float(np.sqrt(abs(float(J))))        # scalar J folds into w = g^2 ...
                if float(J) < 0:
                    raise ValueError('negative scalar J is not supported')
            self.S, Jt = 0, None
        else:
            Jt = _hip.as_f32(J, dev)
            self.S = int(Jt.shape[0])
            Jt = Jt.reshape(self.S, P).contiguous()
        self.Sx = max(self.S, 1)
        self.w = torch.empty((self.Sx, P), dtype=torch.float32, device=dev)
        self.dom = torch.empty((P,), dtype=torch.uint8, device=dev)
        self.rmin, self.rmax, self.z_width = float(rmin), float(rmax), float(z_width)
        _hip.check(_hip.lib().bhn_geom_prepare(
            _hip.ptr(self.coords), _hip.ptr(g), _hip.ptr(dtau), _hip.ptr(Sigma), _hip.ptr(Jt), self.S, P,
            self.rmin, self.rmax, self.z_width, _hip.ptr(self.w), _hip.ptr(self.dom), _hip.stream_ptr(dev)))
        self.device = dev
        # compaction of the static domain mask: the 32-point groups that contain an in-domain point
        ngroups = (P + 31) // 32
        pad = torch.zeros((ngroups * 32,), dtype=torch.uint8, device=dev)
        pad[:P] = self.dom
        active = pad.view(ngroups, 32).any(dim=1)
        self.n_groups_total = ngroups
        if bool(active.all()):
            self.groups, self.n_groups = None, ngroups
        else:
            self.groups = torch.nonzero(active).reshape(-1).to(torch.int32).contiguous()
            self.n_groups = max(int(self.groups.numel()), 1)
            if self.groups.numel() == 0:                      # nothing inside the domain: keep one (masked) group
                self.groups = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.P_eff = self.n_groups * 32                       # points the fused kernels / the tape visit per frame
        # point-level compaction for the fused MLP kernels: when a sizeable part of the samples lies outside the
        # domain, a second copy of the planes holds only the in-domain points (ray-major order kept, padded with
        # masked points to a multiple of 32) plus the ray of every point.  Group-level compaction alone still
        # evaluates every point of a 32-point group that has one in-domain sample (tutorial domain: 61 % visited
        # for 28 % in-domain).
        self.compact = None
        n_in = int(self.dom.sum().item())
        if COMPACT_POINTS and 0 < n_in < COMPACT_BELOW * P:
            idx = torch.nonzero(self.dom).reshape(-1)
            n_pad = (n_in + 31) // 32 * 32

            def take(v, dtype=torch.float32):
                out = torch.zeros((n_pad,), dtype=dtype, device=dev)
                out[:n_in] = v[idx].to(dtype)
                return out
            cw = torch.zeros((self.Sx, n_pad), dtype=torch.float32, device=dev)
            cw[:, :n_in] = self.w[:, idx]
            self.compact = dict(idx=idx, n=n_in, n_pad=n_pad, x=take(self.coords[0]), y=take(self.coords[1]), z=take(self.coords[2]),
                                Omega=take(self.Omega), t_geo=take(self.t_geo), w=cw, dom=take(self.dom, torch.uint8),
                                ray=take(torch.div(torch.arange(P, device=dev), self.G, rounding_mode='floor'), torch.int32))
            self.P_eff = n_pad
            # bhn_geom.ray_span: the most 32-point groups the (consecutive) points of one ray lie in -- 1 or 2 lets the render
            # kernels skip the per-tile combine of ray segments (include/bhnerf_hip.h)
            self.compact['ray_span'] = ray_span(self.compact['ray'][:n_in])

    def c_struct(self):
        """Dense layout (R x G planes): voxel / grid kernels and every caller that indexes points as ray * G + sample."""
        c = self.coords
        return _hip.bhn_geom(self.R, self.G, self.S, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(),
                             self.Omega.data_ptr(), self.t_geo.data_ptr(), self.w.data_ptr(), self.dom.data_ptr(),
                             self.groups.data_ptr() if self.groups is not None else None, self.n_groups, None, 0)

    def c_struct_fused(self):
        """What the fused predictor / render kernels get: the point-compacted planes when they exist."""
        k = self.compact
        if k is None:
            return self.c_struct()
        return _hip.bhn_geom(self.R, self.G, self.S, k['x'].data_ptr(), k['y'].data_ptr(), k['z'].data_ptr(),
                             k['Omega'].data_ptr(), k['t_geo'].data_ptr(), k['w'].data_ptr(), k['dom'].data_ptr(),
                             None, k['n_pad'] // 32, k['ray'].data_ptr(), k['n_pad'], k['ray_span'])

    @property
    def active_fraction(self):
        """Fraction of ray samples inside the supervised domain (static mask)."""
        return float(self.dom.float().mean().item())

    @property
    def visited_fraction(self):
        """Fraction of ray samples the fused kernels evaluate after compaction (points, else 32-point groups)."""
        return self.P_eff / float(self.n_groups_total * 32)


def ray_span(ray):
    """bhn_geom.ray_span of a point-compacted layout: `ray` = the ray of every in-domain point in ray-major order (a 1-D integer
    tensor, runs of equal values); the result is the largest number of consecutive 32-point groups one run lies in."""
    n = int(ray.numel())
    if n == 0:
        return 0
    first = torch.ones(1, dtype=torch.bool, device=ray.device)
    starts = torch.nonzero(torch.cat([first, ray[1:] != ray[:-1]])).reshape(-1)
    ends = torch.cat([starts[1:], torch.tensor([n], device=ray.device)]) - 1
    return int((torch.div(ends, 32, rounding_mode='floor') - torch.div(starts, 32, rounding_mode='floor')).max().item()) + 1


def frame_offsets(t_frames, t_start_obs, t_injection, GM_c3, device):
    """tM0[b] = (t_frames[b]-t_start_obs)/GM_c3 - t_injection in float64 (emission.py:200-201)."""
    t = np.atleast_1d(np.asarray(t_frames, dtype=np.float64))
    tM0 = (t - float(t_start_obs)) / float(GM_c3) - float(t_injection)
    return _hip.h2d_small(np.asarray(tM0, dtype=np.float64), device)      # (pinned staging: no host stall, _hip._PinnedRing)


class FusedPredictor:
    """NeRF_Predictor hyper-parameters + packed weights bound to the fused HIP kernels."""

    def __init__(self, net_depth=4, net_width=128, posenc_deg=3, do_skip=True, scale=1.0, rmin=0.0,
                 rmax=np.inf, z_width=np.inf, mode='bf16', device='cuda'):
        self.mode = _hip.MODES[mode] if isinstance(mode, str) else int(mode)
        self.model = _hip.make_model(net_depth, net_width, posenc_deg, do_skip, scale, rmin, rmax, z_width)
        self.device = torch.device(device)
        lib = _hip.lib()
        self.nparams = int(lib.bhn_param_count(C.byref(self.model)))
        if self.nparams < 0:
            raise _hip.HipError(lib.bhn_last_error().decode())
        n = net_depth + 1
        ko, bo, ind = (C.c_int64 * n)(), (C.c_int64 * n)(), (C.c_int32 * n)()
        _hip.check(lib.bhn_param_layout(C.byref(self.model), ko, bo, ind))
        self.kernel_off, self.bias_off, self.in_dim = list(ko), list(bo), list(ind)
        self.out_dim = [net_width] * net_depth + [1]
        self.packed = torch.empty((int(lib.bhn_packed_bytes(C.byref(self.model), self.mode)),), dtype=torch.uint8,
                                  device=self.device)
        self._ws = None
        # posenc_deg > 4 or net_width > 256: the general layer-by-layer path (csrc/general_mlp.hip)
        self.general = posenc_deg > 4 or net_width > 256
        # queried once, here: torch's device-property query is not safe to call from two host threads at the same time
        self._total_memory = int(torch.cuda.get_device_properties(self.device).total_memory)

    # -- parameters --------------------------------------------------------------------------
    def flatten(self, tree):
        """{'MLP_0': {'Dense_i': {'kernel','bias'}}} -> flat float32 device tensor (flax tree order)."""
        flat = torch.empty((self.nparams,), dtype=torch.float32, device=self.device)
        mlp = tree['MLP_0'] if 'MLP_0' in tree else tree
        for i, (ko, bo, fi, fo) in enumerate(zip(self.kernel_off, self.bias_off, self.in_dim, self.out_dim)):
            k = _hip.as_f32(mlp['Dense_%d' % i]['kernel'], self.device)
            if tuple(k.shape) != (fi, fo):
                raise ValueError('Dense_%d kernel shape %s, expected %s' % (i, tuple(k.shape), (fi, fo)))
            flat[ko:ko + fi * fo] = k.reshape(-1)
            flat[bo:bo + fo] = _hip.as_f32(mlp['Dense_%d' % i]['bias'], self.device).reshape(-1)
        return flat

    def unflatten(self, flat):
        tree = {}
        for i, (ko, bo, fi, fo) in enumerate(zip(self.kernel_off, self.bias_off, self.in_dim, self.out_dim)):
            tree['Dense_%d' % i] = {'kernel': flat[ko:ko + fi * fo].view(fi, fo), 'bias': flat[bo:bo + fo]}
        return {'MLP_0': tree}

    @_on_device
    def pack(self, flat):
        _hip.require_device(flat)
        assert flat.dtype == torch.float32 and flat.numel() == self.nparams and flat.is_contiguous()
        _hip.check(_hip.lib().bhn_pack_weights(C.byref(self.model), self.mode, _hip.ptr(flat), _hip.ptr(self.packed),
                                               _hip.stream_ptr(self.device)))

    # -- fused calls ---------------------------------------------------------------------------
    step_timer = None       # measurement aid (bench.py): an object with fwd_mark(i) / bwd_events() that times the kernels of render_train / render_bwd_tape INSIDE the step loop
    clock_probe = None      # measurement aid (bench.py): an int64 device tensor of 4 * BHN_CLK_SLOTS entries -> bhn_frames.clock_probe

    def _frames(self, tM0):
        assert tM0.dtype == torch.float64 and tM0.is_cuda and tM0.is_contiguous()
        cp = self.clock_probe
        assert cp is None or (cp.dtype == torch.int64 and cp.is_cuda and cp.numel() >= 4 * _hip.BHN_CLK_SLOTS)
        return _hip.bhn_frames(int(tM0.numel()), tM0.data_ptr(), None if cp is None else cp.data_ptr())

    def tape_info(self, groups_per_frame):
        """bhn_tape_info: bytes of tape per 32-point group each kernel of the training step moves, and the layout flags."""
        info = (C.c_int64 * _hip.BHN_TAPE_INFO_N)()
        _hip.check(_hip.lib().bhn_tape_info(C.byref(self.model), self.mode, int(groups_per_frame), info, _hip.BHN_TAPE_INFO_N))
        flags = {k: bool(info[4] & v) for k, v in _hip.TAPE_FLAGS.items()}
        return {'fwd_write': info[0], 'chain_write': info[1], 'chain_read': info[2], 'dw_read': info[3], 'flags': flags,
                'fwd_groups_per_tile': info[5]}

    @_on_device
    def predict(self, geom, tM0):
        """NeRF_Predictor.__call__ (network.py:191-237) -> emission (B, P) float32."""
        B = int(tM0.numel())
        k = geom.compact
        out = torch.empty((B, geom.P if k is None else k['n_pad']), dtype=torch.float32, device=self.device)
        gs, fs = geom.c_struct_fused(), self._frames(tM0)
        _hip.check(_hip.lib().bhn_predict_fwd(C.byref(self.model), self.mode, _hip.ptr(self.packed), C.byref(gs),
                                              C.byref(fs), _hip.ptr(out), _hip.stream_ptr(self.device)))
        if k is None:
            return out
        full = torch.zeros((B, geom.P), dtype=torch.float32, device=self.device)      # outside the domain: 0 (emission.py:370-373)
        full[:, k['idx']] = out[:, :k['n']]
        return full

    @_on_device
    def render(self, geom, tM0, out=None):
        """image_plane_prediction (network.py:373-420) -> images (B, Sx, R) float32."""
        B = int(tM0.numel())
        if out is None:
            out = torch.empty((B, geom.Sx, geom.R), dtype=torch.float32, device=self.device)
        gs, fs = geom.c_struct_fused(), self._frames(tM0)
        _hip.check(_hip.lib().bhn_render_fwd(C.byref(self.model), self.mode, _hip.ptr(self.packed), C.byref(gs),
                                             C.byref(fs), _hip.ptr(out), _hip.stream_ptr(self.device)))
        return out

    @_on_device
    def workspace(self, B, P):
        """Backward workspace (slabs + tape).  Sized for all B frames when that fits under
        ``max_workspace_bytes`` (default 1/4 of the device memory), else for as many frames as fit;
        ``bhn_render_bwd`` then iterates over frame groups."""
        lib = _hip.lib()
        dev = self.device.index or 0
        full = int(lib.bhn_render_bwd_workspace_bytes(C.byref(self.model), self.mode, B, P, dev))
        one = int(lib.bhn_render_bwd_workspace_bytes(C.byref(self.model), self.mode, 1, P, dev))
        if full == 0 or one == 0:
            raise _hip.HipError(lib.bhn_last_error().decode() or 'render_bwd workspace query failed')
        cap = getattr(self, 'max_workspace_bytes', None)
        if cap is None:
            cap = self._total_memory // 4
        # fused paths: at least one frame of tape (what bhn_render_bwd needs).  General path: bhn_render_bwd walks the groups in
        # chunks of any size, so the cap is the cap -- one frame of an 8x512 network on 256 x 256 x 128 rays would be 277 GB
        want = min(full, cap) if self.general else min(full, max(one, cap))
        if self._ws is None or self._ws.numel() < want:
            self._ws = None
            self._ws = torch.empty((want,), dtype=torch.uint8, device=self.device)
            self._t8_calibrated = False        # (the 8-bit tape's scales live in the workspace)
        return self._ws

    def recalibrate(self):
        """8-bit tape mode: take the tape scales of the next backward call from that call itself (it runs its delta chain
        twice) instead of from the call before it.  Done automatically on the first backward call on a new workspace."""
        self._t8_calibrated = False

    def _bwd_mode(self):
        if self.mode != _hip.BHN_BF16_T8 or getattr(self, '_t8_calibrated', False):
            return self.mode
        return self.mode | _hip.BHN_T8_CALIBRATE

    def _bwd_done(self, mode):
        """After a backward call has been ENQUEUED without error: a calibrating call need not be repeated."""
        if mode & _hip.BHN_T8_CALIBRATE:
            self._t8_calibrated = True

    @_on_device
    def render_bwd(self, geom, tM0, dimages, out=None):
        """d loss / d params given d loss / d images (B,Sx,R): the reverse of ``render``."""
        assert dimages.dtype == torch.float32 and dimages.is_contiguous() and dimages.is_cuda
        if out is None:
            out = torch.empty((self.nparams,), dtype=torch.float32, device=self.device)
        ws = self.workspace(int(tM0.numel()), geom.P_eff)
        gs, fs = geom.c_struct_fused(), self._frames(tM0)
        mode = self._bwd_mode()
        _hip.check(_hip.lib().bhn_render_bwd(C.byref(self.model), mode, _hip.ptr(self.packed), C.byref(gs),
                                             C.byref(fs), _hip.ptr(dimages), _hip.ptr(out), _hip.ptr(ws), ws.numel(),
                                             _hip.stream_ptr(self.device)))
        self._bwd_done(mode)
        return out


    def fits_tape(self, B, P):
        """True when the workspace can hold the tape of all B frames (the recorded-tape fast path)."""
        lib = _hip.lib()
        full = int(lib.bhn_render_bwd_workspace_bytes(C.byref(self.model), self.mode, B, P, self.device.index or 0))
        return 0 < full <= self.workspace(B, P).numel()

    def tape_group(self, B, P):
        """Largest number of frames (<= B) whose tape fits the workspace cap, 0 if not even one does: a training
        step whose loss is a sum of per-frame terms then runs frame group by frame group on the recorded-tape
        path instead of recomputing the forward (`bhn_render_bwd`)."""
        lib = _hip.lib()
        dev = self.device.index or 0
        cap = getattr(self, 'max_workspace_bytes', None)
        if cap is None:
            cap = self._total_memory // 4
        for nb in range(int(B), 0, -1):
            need = int(lib.bhn_render_bwd_workspace_bytes(C.byref(self.model), self.mode, nb, P, dev))
            if 0 < need <= cap:
                return nb
        return 0

    @_on_device
    def render_train(self, geom, tM0, out=None):
        """Training forward: images (B,Sx,R) + tape recorded in the workspace (see render_bwd_tape)."""
        B = int(tM0.numel())
        if out is None:
            out = torch.empty((B, geom.Sx, geom.R), dtype=torch.float32, device=self.device)
        ws = self.workspace(B, geom.P_eff)
        gs, fs = geom.c_struct_fused(), self._frames(tM0)
        timer = self.step_timer
        if timer is not None:
            timer.fwd_mark(0)
        _hip.check(_hip.lib().bhn_render_fwd_train(C.byref(self.model), self.mode, _hip.ptr(self.packed), C.byref(gs),
                                                   C.byref(fs), _hip.ptr(out), _hip.ptr(ws), ws.numel(),
                                                   _hip.stream_ptr(self.device)))
        if timer is not None:
            timer.fwd_mark(1)
        return out

    @_on_device
    def render_bwd_tape(self, geom, tM0, dimages, out=None):
        """Gradient from the tape recorded by ``render_train`` (same geom / frames / packed weights)."""
        assert dimages.dtype == torch.float32 and dimages.is_contiguous() and dimages.is_cuda
        if out is None:
            out = torch.empty((self.nparams,), dtype=torch.float32, device=self.device)
        ws = self.workspace(int(tM0.numel()), geom.P_eff)
        gs, fs = geom.c_struct_fused(), self._frames(tM0)
        mode = self._bwd_mode()
        timer = self.step_timer
        if timer is not None:       # bench.py: the caller's HIP events between the kernels of this call (bhn_render_bwd_tape_timed)
            ev, n_ev = timer.bwd_events()
            _hip.check(_hip.lib().bhn_render_bwd_tape_timed(C.byref(self.model), mode, _hip.ptr(self.packed), C.byref(gs),
                                                            C.byref(fs), _hip.ptr(dimages), _hip.ptr(out), _hip.ptr(ws), ws.numel(),
                                                            _hip.stream_ptr(self.device), ev, n_ev))
        else:
            _hip.check(_hip.lib().bhn_render_bwd_tape(C.byref(self.model), mode, _hip.ptr(self.packed), C.byref(gs),
                                                      C.byref(fs), _hip.ptr(dimages), _hip.ptr(out), _hip.ptr(ws), ws.numel(),
                                                      _hip.stream_ptr(self.device)))
        self._bwd_done(mode)
        return out


class RenderFunction(torch.autograd.Function):
    """images = render(params) with the fused HIP forward/backward (torch.autograd glue only)."""

    @staticmethod
    def forward(ctx, flat, predictor, geom, tM0):
        predictor.pack(flat)
        ctx.predictor, ctx.geom, ctx.tM0 = predictor, geom, tM0
        return predictor.render(geom, tM0)

    @staticmethod
    def backward(ctx, dimages):
        return ctx.predictor.render_bwd(ctx.geom, ctx.tM0, dimages.contiguous()), None, None, None


class GridEngine:
    """GRID_Predictor (network.py:254-353) on the HIP kernels bhn_grid_*: the parameter vector IS the (res,res,res)
    voxel grid.  Offers the subset of the FusedPredictor interface the training / rendering drivers use."""

    def __init__(self, grid_res=64, scale=1.0, device='cuda'):
        self.res, self.scale = int(grid_res), float(scale)
        self.device = torch.device(device)
        self.nparams = self.res ** 3
        self.grid = None

    def flatten(self, tree):
        g = tree['grid'] if isinstance(tree, dict) else tree
        g = _hip.as_f32(g, self.device)
        if tuple(g.shape) != (self.res,) * 3:
            raise ValueError('grid shape %s, expected %s' % (tuple(g.shape), (self.res,) * 3))
        return g.reshape(-1).clone()

    def unflatten(self, flat):
        return {'grid': flat.view(self.res, self.res, self.res)}

    @_on_device
    def pack(self, flat):
        _hip.require_device(flat)
        assert flat.dtype == torch.float32 and flat.numel() == self.nparams and flat.is_contiguous()
        self.grid = flat

    def _frames(self, tM0):
        assert tM0.dtype == torch.float64 and tM0.is_cuda and tM0.is_contiguous()
        return _hip.bhn_frames(int(tM0.numel()), tM0.data_ptr())

    @_on_device
    def predict(self, geom, tM0):
        out = torch.empty((int(tM0.numel()), geom.P), dtype=torch.float32, device=self.device)
        gs, fs = geom.c_struct(), self._frames(tM0)
        _hip.check(_hip.lib().bhn_grid_predict_fwd(C.byref(gs), C.byref(fs), _hip.ptr(self.grid), self.res, self.scale,
                                                   _hip.ptr(out), _hip.stream_ptr(self.device)))
        return out

    @_on_device
    def render(self, geom, tM0, out=None):
        B = int(tM0.numel())
        if out is None:
            out = torch.empty((B, geom.Sx, geom.R), dtype=torch.float32, device=self.device)
        gs, fs = geom.c_struct(), self._frames(tM0)
        _hip.check(_hip.lib().bhn_grid_render_fwd(C.byref(gs), C.byref(fs), _hip.ptr(self.grid), self.res, self.scale,
                                                  _hip.ptr(out), _hip.stream_ptr(self.device)))
        return out

    @_on_device
    def render_bwd(self, geom, tM0, dimages, out=None):
        assert dimages.dtype == torch.float32 and dimages.is_contiguous() and dimages.is_cuda
        if out is None:
            out = torch.empty((self.nparams,), dtype=torch.float32, device=self.device)
        gs, fs = geom.c_struct(), self._frames(tM0)
        _hip.check(_hip.lib().bhn_grid_render_bwd(C.byref(gs), C.byref(fs), _hip.ptr(self.grid), self.res, self.scale,
                                                  _hip.ptr(dimages), _hip.ptr(out), _hip.stream_ptr(self.device)))
        return out

    # no tape: the "taped" route of the step driver is simply render + render_bwd
    render_train = render
    render_bwd_tape = render_bwd

    def fits_tape(self, B, P):
        return True

    def tape_group(self, B, P):
        return int(B)


def chi2_image(images, target, sigma, offset, scale, dtype, want_grad=True):
    """loss_fn_image (network.py:476-484) on device -> (loss[1], dimages or None)."""
    code = {'full': 0, 'lc': 1}.get(dtype)
    if code is None:
        raise AttributeError('image dtype ({}) not supported'.format(dtype))
    B, Sx, R = images.shape
    loss = torch.empty((1 + B * Sx,), dtype=torch.float32, device=images.device)     # [total | per-plane terms]
    dimg = torch.empty_like(images) if want_grad else None
    with torch.cuda.device(images.device):
        _hip.check(_hip.lib().bhn_chi2_image(_hip.ptr(images), _hip.ptr(target), _hip.ptr(sigma), _hip.ptr(offset),
                                             float(scale), code, B, Sx, R, _hip.ptr(loss), _hip.ptr(dimg),
                                             _hip.stream_ptr(images.device)))
    return loss[:1], dimg


EHT_DTYPES = {'vis': 0, 'amp': 1, 'cphase': 2}


def _eht_operands(images, A, target, sigma, dtype):
    """Flatten (images, A, target, sigma) to the kernel layout; raises the reference's AttributeErrors
    (network.py:545-562)."""
    code = EHT_DTYPES.get(dtype)
    if code is None:
        raise AttributeError('eht dtype ({}) not supported'.format(dtype))
    vis_ndim = A.ndim - 1
    want = target.ndim + (1 if dtype == 'cphase' else 0)
    if vis_ndim != want:
        raise AttributeError('visibilities (ndim={}) should have {} dimensions as target (ndim={}) for dtype={}'.format(
            vis_ndim, '+1' if dtype == 'cphase' else 'same', target.ndim, dtype))
    R, nvis = int(A.shape[-1]), int(A.shape[-2])
    C_ = int(A.shape[-3]) if dtype == 'cphase' else 1
    N = int(np.prod(target.shape[:-1])) if target.ndim > 1 else 1
    dev = images.device
    img = images.reshape(N, R).to(torch.float32).contiguous()
    A = torch.as_tensor(A, device=dev)
    Ar = torch.view_as_real(A.to(torch.complex64).reshape(N, C_, nvis, R).contiguous())
    tgt = torch.as_tensor(target, device=dev)
    tgt = torch.view_as_real(tgt.to(torch.complex64).reshape(N, nvis).contiguous()) if dtype == 'vis' \
        else tgt.to(torch.float32).reshape(N, nvis).contiguous()
    sig = torch.as_tensor(sigma, device=dev).to(torch.float32).expand(tuple(target.shape)).reshape(N, nvis).contiguous()
    return code, img, Ar, tgt, sig, N, C_, nvis, R


def chi2_eht(images, A, target, sigma, scale, dtype, want_grad=True):
    """loss_fn_eht tail (network.py:541-564) on device -> (loss[1], dimages shaped like images or None)."""
    code, img, Ar, tgt, sig, N, C_, nvis, R = _eht_operands(images, A, target, sigma, dtype)
    dev = img.device
    ws = torch.empty((int(_hip.lib().bhn_chi2_eht_ws_floats(N, C_, nvis, R)),), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dimg = torch.empty_like(img) if want_grad else None
    with torch.cuda.device(dev):
        _hip.check(_hip.lib().bhn_chi2_eht(_hip.ptr(img), _hip.ptr(Ar), _hip.ptr(tgt), _hip.ptr(sig), float(scale), code, N, C_,
                                           nvis, R, _hip.ptr(ws), _hip.ptr(loss), _hip.ptr(dimg), _hip.stream_ptr(dev)))
    return loss, (dimg.reshape(images.shape) if want_grad else None)


class EhtChi2Function(torch.autograd.Function):
    """scale*chi^2 of the visibilities of `images` (differentiable w.r.t. images)."""

    @staticmethod
    def forward(ctx, images, A, target, sigma, scale, dtype):
        loss, dimg = chi2_eht(images, A, target, sigma, scale, dtype, want_grad=True)
        ctx.save_for_backward(dimg)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dimg,) = ctx.saved_tensors
        return dimg * g, None, None, None, None, None


def adam_step(params, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
    with torch.cuda.device(params.device):
        _hip.check(_hip.lib().bhn_adam_step(_hip.ptr(params), _hip.ptr(grads), _hip.ptr(m), _hip.ptr(v), params.numel(),
                                            int(t), float(lr), b1, b2, eps, float(grad_scale),
                                            _hip.stream_ptr(params.device)))


def adam_hyper(t, lr, b1=0.9, b2=0.999):
    """(lr, 1 - b1^t, 1 - b2^t) as float32, computed by the library exactly as bhn_adam_step does (host)."""
    h = (C.c_float * 3)()
    _hip.check(_hip.lib().bhn_adam_hyper(int(t), float(lr), b1, b2, h))
    return np.array(list(h), dtype=np.float32)


def adam_step_dev(params, grads, m, v, hyper_dev, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
    """Adam with lr / bias corrections read from device memory (hyper_dev: 3 float32): no per-step scalar in the launch, so the
    step can sit inside a HIP graph.  Bitwise equal to adam_step given adam_hyper's values."""
    assert hyper_dev.dtype == torch.float32 and hyper_dev.numel() >= 3 and hyper_dev.is_cuda
    with torch.cuda.device(params.device):
        _hip.check(_hip.lib().bhn_adam_step_dev(_hip.ptr(params), _hip.ptr(grads), _hip.ptr(m), _hip.ptr(v), params.numel(),
                                                _hip.ptr(hyper_dev), b1, b2, eps, float(grad_scale), _hip.stream_ptr(params.device)))
