"""ctypes binding of libbhnerf_hip.so (include/bhnerf_hip.h).

PyTorch is used only as the owner of device memory and streams: every call passes raw
``data_ptr()`` addresses and the current HIP stream handle through the C ABI.  There is no CPU
fallback: if the library cannot be loaded, or a call fails, a ``HipError`` is raised.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# BHNERF_HIP_LIB: an alternative build of the library, e.g. the debug build (make debug) for the tools/ scripts
LIB_PATH = os.environ.get('BHNERF_HIP_LIB') or os.path.join(CSRC, 'libbhnerf_hip.so')
DEBUG_SIGNATURES = {        # include/bhnerf_hip_debug.h: only libbhnerf_hip_dbg.so exports these
    'bhn_debug_set_bwd_stages': (C.c_int, [C.c_int32]),
    'bhn_debug_set_fwd_variant': (C.c_int, [C.c_int32]),
    'bhn_debug_read': (C.c_int, [C.c_void_p, C.c_size_t]),
}

ABI_VERSION = 5              # BHN_ABI_VERSION of include/bhnerf_hip.h this binding was written against
BHN_F32, BHN_BF16, BHN_BF16_T8 = 0, 1, 2
BHN_T8_CALIBRATE = 0x100
BHN_CLK_FWD, BHN_CLK_FWD_TRAIN, BHN_CLK_CHAIN, BHN_CLK_DW, BHN_CLK_SLOTS = 0, 1, 2, 3, 4      # kernel slots of bhn_frames.clock_probe
BHN_TAPE_INFO_N = 8
TAPE_FLAGS = {'drop_h1': 1, 'drop_ga': 2, 'ga0_chain': 4, 'fused128': 8, 'drop_hd': 16, 'lbits': 32, 'general': 64}
MODES = {'f32': BHN_F32, 'fp32': BHN_F32, 'float32': BHN_F32, 'bf16': BHN_BF16, 'bfloat16': BHN_BF16,
         'bf16_t8': BHN_BF16_T8}            # bf16 arithmetic, 8-bit (e4m3) backward tape: include/bhnerf_hip.h


class HipError(RuntimeError):
    pass


class bhn_model(C.Structure):
    _fields_ = [('net_depth', C.c_int32), ('net_width', C.c_int32), ('posenc_deg', C.c_int32),
                ('do_skip', C.c_int32), ('scale', C.c_float), ('rmin', C.c_float), ('rmax', C.c_float),
                ('z_width', C.c_float)]


class bhn_geom(C.Structure):
    _fields_ = [('R', C.c_int64), ('G', C.c_int64), ('S', C.c_int32), ('x', C.c_void_p), ('y', C.c_void_p),
                ('z', C.c_void_p), ('Omega', C.c_void_p), ('t_geo', C.c_void_p), ('w', C.c_void_p),
                ('dom', C.c_void_p), ('groups', C.c_void_p), ('n_groups', C.c_int64), ('ray_idx', C.c_void_p),
                ('n_points', C.c_int64), ('ray_span', C.c_int32)]


class bhn_frames(C.Structure):
    _fields_ = [('B', C.c_int32), ('tM0', C.c_void_p), ('clock_probe', C.c_void_p)]      # clock_probe: NULL unless bench.py measures the kernels' clocks


_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_MP, _GP, _FP = C.POINTER(bhn_model), C.POINTER(bhn_geom), C.POINTER(bhn_frames)

# name -> (restype, argtypes); must list every symbol include/bhnerf_hip.h declares
SIGNATURES = {
    'bhn_version': (C.c_int, []),
    'bhn_last_error': (C.c_char_p, []),
    'bhn_param_count': (_I64, [_MP]),
    'bhn_param_layout': (C.c_int, [_MP, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I32)]),
    'bhn_geom_prepare': (C.c_int, [_P, _P, _P, _P, _P, _I32, _I64, _F, _F, _F, _P, _P, _P]),
    'bhn_radiative_transfer_fwd': (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    'bhn_radiative_transfer_bwd': (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    'bhn_packed_bytes': (_SZ, [_MP, _I32]),
    'bhn_pack_weights': (C.c_int, [_MP, _I32, _P, _P, _P]),
    'bhn_predict_fwd': (C.c_int, [_MP, _I32, _P, _GP, _FP, _P, _P]),
    'bhn_render_fwd': (C.c_int, [_MP, _I32, _P, _GP, _FP, _P, _P]),
    'bhn_render_bwd_workspace_bytes': (_SZ, [_MP, _I32, _I32, _I64, _I32]),
    'bhn_render_bwd': (C.c_int, [_MP, _I32, _P, _GP, _FP, _P, _P, _P, _SZ, _P]),
    'bhn_render_fwd_train': (C.c_int, [_MP, _I32, _P, _GP, _FP, _P, _P, _SZ, _P]),
    'bhn_render_bwd_tape': (C.c_int, [_MP, _I32, _P, _GP, _FP, _P, _P, _P, _SZ, _P]),
    'bhn_chi2_image': (C.c_int, [_P, _P, _P, _P, _F, _I32, _I32, _I32, _I64, _P, _P, _P]),
    'bhn_chi2_eht_ws_floats': (_SZ, [_I32, _I32, _I32, _I64]),
    'bhn_chi2_eht': (C.c_int, [_P, _P, _P, _P, _F, _I32, _I32, _I32, _I32, _I64, _P, _P, _P, _P]),
    'bhn_voxel_render_fwd': (C.c_int, [_GP, _FP, _P, _I32, _I32, _I32, _I64, C.POINTER(C.c_float), _P, _P]),
    'bhn_trilinear': (C.c_int, [_P, _I64, _P, _I32, _I32, _I32, C.POINTER(C.c_float), _P, _P]),
    'bhn_grid_predict_fwd': (C.c_int, [_GP, _FP, _P, _I32, _F, _P, _P]),
    'bhn_grid_render_fwd': (C.c_int, [_GP, _FP, _P, _I32, _F, _P, _P]),
    'bhn_grid_render_bwd': (C.c_int, [_GP, _FP, _P, _I32, _F, _P, _P, _P]),
    'bhn_adam_step': (C.c_int, [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _P]),
    'bhn_adam_hyper': (C.c_int, [_I64, _F, _F, _F, C.POINTER(C.c_float)]),
    'bhn_adam_step_dev': (C.c_int, [_P, _P, _P, _P, _I64, _P, _F, _F, _F, _F, _P]),
    'bhn_render_bwd_tape_timed': (C.c_int, [_MP, _I32, _P, _GP, _FP, _P, _P, _P, _SZ, _P, C.POINTER(C.c_void_p), _I32]),
    'bhn_render_bwd_tape_kernel_name': (C.c_char_p, [_I32]),
    'bhn_render_bwd_tape_kernel_name_for': (C.c_char_p, [_MP, _I32, _I32]),
    'bhn_tape_info': (C.c_int, [_MP, _I32, _I64, C.POINTER(_I64), _I32]),
    'bhn_mfma_probe': (C.c_int, [_I32, _I32, _P, _P, _P]),
    'bhn_selftest': (C.c_int, [C.POINTER(_I32), _P, _SZ]),
}

_lib = None


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(['make', '-C', CSRC, '-j4'], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise HipError('building libbhnerf_hip.so failed (make -C %s)' % CSRC)
    return LIB_PATH


def lib():
    """The loaded library; raises HipError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(or make -C bhnerf_amd/csrc). There is no CPU fallback.' % LIB_PATH)
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise HipError('cannot load %s: %s' % (LIB_PATH, exc))
        try:
            handle.bhn_version.restype = C.c_int
            have = int(handle.bhn_version())
        except AttributeError:
            have = -1
        if have != ABI_VERSION:      # a stale build would fail later, at a symbol lookup or as "bad mode"
            raise HipError('%s implements ABI version %d, this package binds version %d: rebuild it (make -C bhnerf_amd/csrc)'
                           % (LIB_PATH, have, ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in DEBUG_SIGNATURES.items():
            if hasattr(handle, name):
                fn = getattr(handle, name)
                fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise HipError('libbhnerf_hip: %s (code %d)' % (lib().bhn_last_error().decode(), rc))


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipError('device tensor required (the HIP path has no CPU fallback)')


def as_f32(x, device):
    """Contiguous float32 device tensor from array-like input (host->device copy if needed)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)), device=device)


class _PinnedRing:
    """Small host -> device copies that do not stall the host: a copy from PAGEABLE memory is synchronous on ROCm (the call
    returns only after the stream has reached it, i.e. after every kernel launched before it: one such copy per training step
    -- the frame offsets, the frame indices -- made the host wait for the GPU at every step and then pay the launch latency of
    the next step's ~25 kernels in the open: 0.1-0.5 ms per step).  A ring of pinned staging slots; a slot is re-used only
    after the copy issued from it has executed."""

    def __init__(self, device, slots=32, nbytes=4096):
        self.device, self.nbytes = device, nbytes
        self.slots = [dict(buf=torch.zeros(nbytes, dtype=torch.uint8).pin_memory(), done=None) for _ in range(slots)]
        for sl in self.slots:
            sl['np'] = sl['buf'].numpy()
        self.i = 0

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        n = arr.nbytes
        if n == 0 or n > self.nbytes:
            return torch.as_tensor(arr, device=self.device)
        sl = self.slots[self.i]
        self.i = (self.i + 1) % len(self.slots)
        if sl['done'] is not None:
            sl['done'].synchronize()                 # (waits only when the GPU is a whole ring behind the host)
        else:
            sl['done'] = torch.cuda.Event()
        sl['np'][:n] = arr.reshape(-1).view(np.uint8)
        out = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0].reshape(-1)).dtype, device=self.device)
        out.view(torch.uint8).reshape(-1).copy_(sl['buf'][:n], non_blocking=True)
        sl['done'].record(torch.cuda.current_stream(self.device))
        return out


_rings = {}


def h2d_small(arr, device):
    """Device tensor with the contents (shape, dtype) of a small NumPy array, through the device's pinned staging ring."""
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.as_tensor(np.ascontiguousarray(arr), device=device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ring = _rings.get(key)
    if ring is None:
        ring = _rings[key] = _PinnedRing(torch.device('cuda', key[1]))
    return ring.to_device(arr)


def make_model(net_depth, net_width, posenc_deg, do_skip, scale, rmin, rmax, z_width):
    big = 3.0e38
    clip = lambda v: float(min(max(v, -big), big))
    return bhn_model(int(net_depth), int(net_width), int(posenc_deg), int(bool(do_skip)), clip(scale), clip(rmin),
                     clip(rmax), clip(z_width))


def selftest():
    res = (C.c_int32 * 8)()
    scratch = torch.zeros((16384,), dtype=torch.uint8, device='cuda')
    check(lib().bhn_selftest(res, ptr(scratch), scratch.numel()))
    return list(res), lib().bhn_last_error().decode()
