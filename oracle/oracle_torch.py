"""PyTorch-CPU restatement of the hot path (test oracle + timed CPU baseline).

Same math as ``oracle_np`` but differentiable (torch.autograd stands in for jax.grad,
network.py:617) and multi-threaded (5 un-fused GEMMs + elementwise + autograd + Adam:
the analogue of what XLA-CPU executes for the reference).  NOT shipped code.
"""
import math

import torch

from . import oracle_np as onp


def warp(coords, Omega, t_frames, t_start_obs, t_geos, t_injection, GM_c3):
    """emission.py:200-210 for rot_axis=z, frames on the leading axis.
    coords (3,*sp); Omega,t_geos (*sp) ; t_frames (b,) -> (b,*sp,3) with NaN pre-injection."""
    sp = coords.shape[1:]
    tf = t_frames.reshape((-1,) + (1,) * len(sp))
    t_M = (tf - t_start_obs) / GM_c3 + t_geos - t_injection
    theta = t_M * Omega
    theta = torch.where(t_M < 0.0, torch.full_like(theta, float('nan')), theta)
    # rot_z(-theta) applied to (x,y,z) == utils.rotation_matrix([0,0,1], -theta) @ coords
    c, s = torch.cos(theta), torch.sin(theta)
    x, y, z = coords[0], coords[1], coords[2]
    return torch.stack([c * x + s * y, -s * x + c * y, z + 0.0 * theta], dim=-1)


def posenc(x, deg):
    """network.py:98-122."""
    if deg == 0:
        return x
    scales = torch.tensor([2.0 ** i for i in range(deg)], dtype=x.dtype, device=x.device)
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], -1)
    feat = torch.sin(torch.remainder(torch.cat([xb, xb + 0.5 * math.pi], dim=-1), 100.0 * math.pi))
    return torch.cat([x, feat], dim=-1)


def mlp(kernels, biases, x, net_depth, do_skip=True):
    """network.py:49-62."""
    inputs = x
    skip_layer = net_depth // 2 if do_skip else None
    for i in range(net_depth):
        x = torch.relu(x @ kernels[i] + biases[i])
        if do_skip and i % skip_layer == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    return x @ kernels[net_depth] + biases[net_depth]


def predictor(kernels, biases, t_frames, coords, Omega, t_start_obs, t_geos, t_injection, GM_c3,
              scale, rmin, rmax, z_width, posenc_deg=3, net_depth=4, do_skip=True):
    """network.py:219-233 -> emission (b,*sp)."""
    warped = warp(coords, Omega, t_frames, t_start_obs, t_geos, t_injection, GM_c3)
    valid = torch.isfinite(warped)
    net_in = torch.where(valid, warped, torch.zeros_like(warped))
    out = mlp(kernels, biases, posenc(net_in / scale, posenc_deg), net_depth, do_skip)
    e = torch.sigmoid(out[..., 0] - 10.0)
    r_sq = (coords ** 2).sum(0)
    keep = ~((r_sq < rmin ** 2) | (r_sq > rmax ** 2) | (coords[2].abs() > z_width))
    e = torch.where(keep, e, torch.zeros_like(e))
    return torch.where(valid[..., 0], e, torch.zeros_like(e))


def render(emission, J, g, dtau, Sigma):
    """network.py:415-419 + kgeo.py:618-621.  J None/scalar -> (b,H,W); J (S,*sp) -> (b,S,H,W)
    (the reference additionally squeezes unit axes, network.py:418; callers apply that)."""
    w = g ** 2 * dtau * Sigma
    if J is None:
        return (emission * w).sum(-1)
    return (emission[:, None] * (J * w)[None]).sum(-1)


def loss_image(images, target, sigma, offset, scale, dtype):
    """network.py:476-484."""
    if dtype == 'full':
        return scale * (((images - target - offset) / sigma).abs() ** 2).sum()
    if dtype == 'lc':
        lc = images.sum(dim=(-1, -2))
        return scale * (((lc - target - offset) / sigma).abs() ** 2).sum()
    raise AttributeError('image dtype ({}) not supported'.format(dtype))


def tree_to_lists(tree, dtype=torch.float64, requires_grad=False):
    n = len(tree['MLP_0'])
    ks = [torch.tensor(tree['MLP_0']['Dense_%d' % i]['kernel'], dtype=dtype, requires_grad=requires_grad)
          for i in range(n)]
    bs = [torch.tensor(tree['MLP_0']['Dense_%d' % i]['bias'], dtype=dtype, requires_grad=requires_grad)
          for i in range(n)]
    return ks, bs


class CpuTrainer:
    """gradient_step_image (network.py:566-622) on the CPU: value_and_grad + Adam with the
    linear-decay schedule of network.py:173-174.  Used by tests (float64) and as the timed
    CPU baseline in bench.py (float32, all host threads)."""

    def __init__(self, kernels, biases, geom, hp, num_iters=5000, lr_init=1e-4, lr_final=1e-6):
        self.k = [k.clone().requires_grad_(True) for k in kernels]
        self.b = [b.clone().requires_grad_(True) for b in biases]
        self.geom, self.hp = geom, hp
        self.num_iters, self.lr_init, self.lr_final = num_iters, lr_init, lr_final
        self.m = [torch.zeros_like(p) for p in self.k + self.b]
        self.v = [torch.zeros_like(p) for p in self.k + self.b]
        self.count = 0

    def forward(self, t_frames):
        G, hp = self.geom, self.hp
        e = predictor(self.k, self.b, t_frames, G['coords'], G['Omega'], G['t_start_obs'], G['t_geos'],
                      G['t_injection'], hp['GM_c3'], hp['scale'], hp['rmin'], hp['rmax'], hp['z_width'],
                      hp.get('posenc_deg', 3), hp['net_depth'], hp.get('do_skip', True))
        return render(e, G.get('J'), G['g'], G['dtau'], G['Sigma'])

    def loss_and_grad(self, t_frames, target, sigma, offset, scale, dtype):
        for p in self.k + self.b:
            p.grad = None
        images = self.forward(t_frames)
        loss = loss_image(images, target, sigma, offset, scale, dtype)
        loss.backward()
        return loss.detach(), images.detach(), [p.grad for p in self.k + self.b]

    def step(self, t_frames, target, sigma, offset, scale=1.0, dtype='full', grad_div=1.0):
        loss, images, grads = self.loss_and_grad(t_frames, target, sigma, offset, scale, dtype)
        self.apply(grads, grad_div)
        return loss, images

    def apply(self, grads, grad_div=1.0):
        """One Adam update with the linearly decayed learning rate (split out of step() for the stale-gradient test)."""
        lr = onp.linear_lr(self.count, self.lr_init, self.lr_final, self.num_iters)
        self.count += 1
        t = self.count
        with torch.no_grad():
            for p, g, m, v in zip(self.k + self.b, grads, self.m, self.v):
                g = g / grad_div
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                mhat = m / (1.0 - 0.9 ** t)
                vhat = v / (1.0 - 0.999 ** t)
                p.sub_(lr * mhat / (vhat.sqrt() + 1e-8))


def grid_loss_and_grad(grid, t_frames, geom, hp, target, sigma):
    """chi^2 of the GRID_Predictor render (network.py:306-353 + kgeo.py:621 + network.py:476-480) and its gradient
    w.r.t. the grid by torch.autograd (float64).  geom / hp as for CpuTrainer; `grid` (res,res,res) array."""
    import numpy as np
    g = torch.tensor(np.asarray(grid), dtype=torch.float64, requires_grad=True)
    res = g.shape[0]
    w = warp(geom['coords'], geom['Omega'], t_frames, geom['t_start_obs'], geom['t_geos'], geom['t_injection'], hp['GM_c3'])
    valid = torch.isfinite(w)
    u = torch.where(valid, w, torch.zeros_like(w))
    idx = (u + hp['scale']) / (2.0 * hp['scale']) * (res - 1.0)
    ix, iy, iz = idx[..., 0], idx[..., 1], idx[..., 2]
    inside = (ix >= 0) & (ix <= res - 1) & (iy >= 0) & (iy <= res - 1) & (iz >= 0) & (iz <= res - 1)
    z = torch.zeros_like(ix)
    ixc, iyc, izc = torch.where(inside, ix, z), torch.where(inside, iy, z), torch.where(inside, iz, z)
    hi = max(res - 2, 0)
    x0 = torch.clamp(torch.floor(ixc).long(), max=hi); y0 = torch.clamp(torch.floor(iyc).long(), max=hi); z0 = torch.clamp(torch.floor(izc).long(), max=hi)
    x1, y1, z1 = torch.clamp(x0 + 1, max=res - 1), torch.clamp(y0 + 1, max=res - 1), torch.clamp(z0 + 1, max=res - 1)
    tx, ty, tz = ixc - x0, iyc - y0, izc - z0
    val = ((g[x0, y0, z0] * (1 - tz) + g[x0, y0, z1] * tz) * (1 - ty) + (g[x0, y1, z0] * (1 - tz) + g[x0, y1, z1] * tz) * ty) * (1 - tx) + \
          ((g[x1, y0, z0] * (1 - tz) + g[x1, y0, z1] * tz) * (1 - ty) + (g[x1, y1, z0] * (1 - tz) + g[x1, y1, z1] * tz) * ty) * tx
    val = torch.where(inside, val, z)
    e = torch.sigmoid(val - 10.0)
    c = geom['coords']
    r2 = (c ** 2).sum(0)
    dom = (r2 >= hp['rmin'] ** 2) & (r2 <= hp['rmax'] ** 2) & (c[2].abs() <= hp['z_width'])
    e = torch.where(dom & valid[..., 0], e, torch.zeros_like(e))
    img = (geom['g'] ** 2 * e * geom['dtau'] * geom['Sigma']).sum(-1)
    loss = (((img - target) / sigma) ** 2).sum()
    loss.backward()
    return loss.detach(), img.detach(), g.grad.detach()
