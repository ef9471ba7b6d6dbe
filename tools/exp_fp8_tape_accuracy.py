"""What an 8-bit dW-operand tape would do to the gradient, emulated on the CPU -- no kernel involved.

The bf16 kernels round the layer inputs h_l and the pre-activation gradients gA_l to bf16, form dW_l = sum_p gA_l[p]^T h_l[p] in
f32, and keep everything else as it is.  An 8-bit tape would store exactly those two operands as e4m3 / e5m2 (h_l >= 0 as it
is; gA_l with one power-of-two scale per 32-point tape group, the granularity of a block-scaled MFMA) while the forward and
the delta chain stay bf16.  This script runs the float64 oracle on a config-2-like problem, then forms every dW_l three ways
from the SAME float64 h_l / gA_l -- operands rounded to bf16, to e4m3, to e5m2 (gA) + e4m3 (h) -- and prints the relative L2
error of each against the exact float64 gradient.  (The bf16 column is the operand-rounding part of the bf16 mode's error: the
full mode also rounds inside the chain.)
    python tools/exp_fp8_tape_accuracy.py [rays_per_side] [frames]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bhnerf_amd import synthetic, constants
from oracle import oracle_np as onp, oracle_torch as ot

torch.set_num_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G, depth, width = 64, 4, 256
geo = synthetic.synthetic_geodesics(N, N, G, fov_M=16.0, inc_deg=60.0, seed=0)
GM = constants.GM_c3('hr')
t64 = lambda v: torch.tensor(np.asarray(v, dtype=np.float64))
rng = np.random.default_rng(1)
tree = onp.he_uniform_params(rng, depth, width, 21)
ks, bs = ot.tree_to_lists(tree, torch.float64)
t_frames = t64(np.linspace(0.0, 1.0, 64)[:B * 8:8])
target = t64(synthetic.hotspot_movie(geo, t_frames.numpy(), GM))

# ---- float64 forward with every layer input kept, manual backward with every gA kept -----------------------------------
coords, Om, tg = t64(geo['coords']), t64(geo['Omega']), t64(geo['t_geos'])
warped = ot.warp(coords, Om, t_frames, 0.0, tg, float(geo['t_injection']), GM)
valid = torch.isfinite(warped)
x0 = ot.posenc(torch.where(valid, warped, torch.zeros_like(warped)) / 8.0, 3).reshape(-1, 21)       # (points, 21)
h = [x0]
x = x0
for i in range(depth):
    a = x @ ks[i] + bs[i]
    x = torch.relu(a)
    if i % (depth // 2) == 0 and i > 0:
        x = torch.cat([x, x0], dim=-1)
    h.append(x)                                          # h[i+1]: input of layer i+1
out = (h[depth] @ ks[depth] + bs[depth])[:, 0]
e = torch.sigmoid(out - 10.0) * valid[..., 0].reshape(-1)
w = t64(geo['g'] ** 2 * geo['dtau'] * geo['Sigma'])
img = (e.reshape(B, N, N, G) * w).sum(-1)
dimg = 2.0 * (img - target)                              # d chi^2 / d image, sigma = 1
dE = (dimg[..., None] * w).reshape(-1)
dout = dE * e * (1.0 - e)
gA = [None] * (depth + 1)
gA[depth] = dout[:, None]                                # output layer's pre-activation gradient
gx = gA[depth] @ ks[depth].T                             # gradient w.r.t. h[depth]
for i in range(depth - 1, -1, -1):
    gx = gx[:, :width]                                   # (the skip part of a concat input has no consumer upstream)
    gA[i] = gx * (h[i + 1][:, :width] > 0)
    if i > 0:
        gx = gA[i] @ ks[i].T
P = x0.shape[0]
print('%d points (%d frames x %d^2 rays x %d samples), 4x%d network, float64 reference' % (P, B, N, G, width))


def rnd(x, dt):
    return x.to(torch.float32).to(dt).to(torch.float64)


def group_scaled(x, dt, fmax):
    """One power-of-two scale per 32-point group (all rows of the group share it): x / s -> dt -> * s."""
    g = x.reshape(-1, 32, x.shape[-1])
    m = g.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-300)
    s = torch.exp2(torch.ceil(torch.log2(m / fmax)))
    return (rnd(g / s, dt) * s).reshape(x.shape)


E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2
tot = {k: [0.0, 0.0] for k in ('bf16', 'e4m3 / e4m3', 'e5m2 (gA) / e4m3 (h)')}
print('%-10s %12s %14s %22s' % ('layer', 'bf16', 'e4m3 / e4m3', 'e5m2 (gA) / e4m3 (h)'))
for l in range(depth + 1):
    exact = gA[l].T @ h[l]
    row = []
    for name, ga_q, h_q in (('bf16', rnd(gA[l], torch.bfloat16), rnd(h[l], torch.bfloat16)),
                            ('e4m3 / e4m3', group_scaled(gA[l], E4, 448.0), rnd(h[l], E4)),
                            ('e5m2 (gA) / e4m3 (h)', group_scaled(gA[l], E5, 57344.0), rnd(h[l], E4))):
        d = ga_q.T @ h_q - exact
        row.append(float(d.norm() / exact.norm()))
        tot[name][0] += float(d.norm() ** 2); tot[name][1] += float(exact.norm() ** 2)
    print('dW_%-7d %12.2e %14.2e %22.2e' % (l, *row))
print('%-10s %12.2e %14.2e %22.2e' % ('all layers', *[np.sqrt(v[0] / v[1]) for v in tot.values()]))


# ---- one scale per LAYER (what an unscaled 8-bit MFMA can accumulate without touching the accumulator between groups) -----------
def layer_scaled(x, dt, fmax):
    s = torch.exp2(torch.ceil(torch.log2(x.abs().amax().clamp_min(1e-300) / fmax)))
    return rnd(x / s, dt) * s


print('\none power-of-two scale per layer instead of per 32-point group:')
tot2 = {k: [0.0, 0.0] for k in ('e4m3 / e4m3', 'e5m2 (gA) / e4m3 (h)', 'e5m2 / e5m2')}
print('%-10s %14s %22s %14s   flushed gA (e4m3)' % ('layer', *tot2.keys()))
for l in range(depth + 1):
    exact = gA[l].T @ h[l]
    row = []
    for name, ga_q, h_q in (('e4m3 / e4m3', layer_scaled(gA[l], E4, 448.0), layer_scaled(h[l], E4, 448.0)),
                            ('e5m2 (gA) / e4m3 (h)', layer_scaled(gA[l], E5, 57344.0), layer_scaled(h[l], E4, 448.0)),
                            ('e5m2 / e5m2', layer_scaled(gA[l], E5, 57344.0), layer_scaled(h[l], E5, 57344.0))):
        d = ga_q.T @ h_q - exact
        row.append(float(d.norm() / exact.norm()))
        tot2[name][0] += float(d.norm() ** 2); tot2[name][1] += float(exact.norm() ** 2)
    q = layer_scaled(gA[l], E4, 448.0)
    print('dW_%-7d %14.2e %22.2e %14.2e   %.3f of the nonzero entries' % (l, *row, float(((q == 0) & (gA[l] != 0)).sum()) / max(float((gA[l] != 0).sum()), 1)))
print('%-10s %14.2e %22.2e %14.2e' % ('all layers', *[np.sqrt(v[0] / v[1]) for v in tot2.values()]))
