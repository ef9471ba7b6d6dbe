"""bhnerf.kgeo hot-path member: the radiative-transfer ray integral (kgeo.py:595-622).

The reference spells it ``radiative_trasfer``; both spellings are exported.  Device tensors go
through the HIP kernel ``bhn_radiative_transfer_fwd``/``_bwd`` (differentiable w.r.t. emission);
NumPy inputs follow the reference's own NumPy path (``use_jax=False``).  The GR pre-compute part of
bhnerf/kgeo.py (kgeo.py:6-593) is out of scope.
"""
import numpy as np
import torch

from . import _hip, utils


def _plane_shape(emission, *vs):
    """Shape (*spatial, G) of the per-ray factors: the first array-valued factor decides, else the
    trailing (H, W, G) axes of the emission."""
    for v in vs:
        if np.ndim(v) > 0:
            return tuple(v.shape)
    return tuple(emission.shape[-3:]) if emission.ndim >= 3 else tuple(emission.shape)


class _RadiativeTransfer(torch.autograd.Function):
    """img[n, r] = sum_k g^2 e[n, r, k] dtau Sigma through the C ABI (fwd and bwd kernels)."""

    @staticmethod
    def forward(ctx, emission, g, dtau, Sigma):
        full = tuple(g.shape)
        G, R = full[-1], int(np.prod(full[:-1]))
        lead = tuple(emission.shape[:emission.ndim - len(full)])
        N = int(np.prod(lead)) if lead else 1
        out = torch.empty(lead + full[:-1], dtype=torch.float32, device=emission.device)
        _hip.check(_hip.lib().bhn_radiative_transfer_fwd(
            _hip.ptr(emission), _hip.ptr(g), _hip.ptr(dtau), _hip.ptr(Sigma), _hip.ptr(out), N, R, G,
            _hip.stream_ptr(emission.device)))
        ctx.save_for_backward(g, dtau, Sigma)
        ctx.dims = (tuple(emission.shape), N, R, G)
        return out

    @staticmethod
    def backward(ctx, dimg):
        g, dtau, Sigma = ctx.saved_tensors
        eshape, N, R, G = ctx.dims
        de = torch.empty(eshape, dtype=torch.float32, device=dimg.device)
        _hip.check(_hip.lib().bhn_radiative_transfer_bwd(
            _hip.ptr(dimg.contiguous()), _hip.ptr(g), _hip.ptr(dtau), _hip.ptr(Sigma), _hip.ptr(de), N, R, G,
            _hip.stream_ptr(dimg.device)))
        return de, None, None, None


def radiative_trasfer(emission, g, dtau, Sigma, use_jax=False):
    """stokes = sum_k g^2 * emission * dtau * Sigma over the last (geodesic-sample) axis."""
    if isinstance(emission, torch.Tensor):
        _hip.require_device(emission)
        full = _plane_shape(emission, g, dtau, Sigma)
        dev = emission.device
        planes = [_hip.as_f32(v, dev).expand(full).contiguous() for v in (g, dtau, Sigma)]
        return _RadiativeTransfer.apply(emission.to(torch.float32).contiguous(), *planes)
    nd = np.ndim(emission)
    g, dtau, Sigma = (utils.expand_dims(v, nd) for v in (g, dtau, Sigma))
    return (g ** 2 * emission * dtau * Sigma).sum(axis=-1)


radiative_transfer = radiative_trasfer
