import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return load


def golden_tree(g):
    """{'MLP_0': {'Dense_i': {'kernel','bias'}}} from a g5_* fixture."""
    n = len([k for k in g if k.startswith('kernel')])
    return {'MLP_0': {'Dense_%d' % i: {'kernel': g['kernel%d' % i], 'bias': g['bias%d' % i]} for i in range(n)}}


# ---------------------------------------------------------------------------------------------------------------
# Where f32 and f64 may LEGITIMATELY disagree (used by the GPU parity tests to tighten every other comparison):
# ---------------------------------------------------------------------------------------------------------------
F32_EPS = 2.0 ** -23


def mask_tie_points(g, t_frames=None):
    """Boolean (b, *spatial, G) array of the (frame, point) pairs whose domain / injection mask is decided within f32
    rounding in the float64 reference: r^2 within a few f32 ulps of rmin^2 or rmax^2, |z| of z_width (emission.py:370-373;
    the device gets f32-rounded coordinates), or t_M of 0 (emission.py:204-205; t_geos is f32 on the device).  Everywhere
    else the kernel's masks must equal the reference's exactly."""
    hp = g['hparams']
    c = np.asarray(g['coords'], dtype=np.float64)
    r2 = (c ** 2).sum(0)
    near = lambda v, b: np.isfinite(b) & (np.abs(v - b) <= 8 * F32_EPS * np.maximum(np.abs(b), 1e-30))
    static = near(r2, hp[1] ** 2) | near(r2, hp[2] ** 2) | near(np.abs(c[2]), hp[3])
    tf = np.atleast_1d(np.asarray(g['t_frames'] if t_frames is None else t_frames, dtype=np.float64))
    from oracle import oracle_np as onp
    tM = (tf.reshape((-1,) + (1,) * g['t_geos'].ndim) - float(g['t_start_obs'])) / onp.GM_C3_SGRA_HR + g['t_geos'] - float(g['t_injection'])
    dyn = np.abs(tM) <= 4 * F32_EPS * (np.abs(g['t_geos']) + abs(float(g['t_injection'])) + 1.0)
    return static[None] | dyn


def relu_tie_count(g, rel=64 * F32_EPS, return_points=False):
    """Number of (frame, point, hidden unit) pre-activations of the float64 reference forward that lie within f32
    rounding of zero: |a| <= 64 eps_f32 (|x| @ |W| + |b|) -- the error band of a 32..288-term f32 dot product whose inputs
    carry the rounding of the layers before (4 ulp of the RESULT would ignore the cancellation that makes it small).
    There f32 and f64 may disagree on relu' and the parameter gradient, which is discontinuous at that point, changes by
    that one point's contribution.  return_points: also the boolean (*spatial, G) array of the ray samples that have such
    a pre-activation in any frame (the tests adjudicate a tie by taking exactly these samples out of the problem)."""
    from oracle import oracle_np as onp
    hp = g['hparams']
    tree = golden_tree(g)['MLP_0']
    depth = int(hp[5])
    warped = onp.velocity_warp_coords(g['coords'], g['Omega'], g['t_frames'], float(g['t_start_obs']), g['t_geos'],
                                      float(g['t_injection']), GM_c3=onp.GM_C3_SGRA_HR)
    valid = np.isfinite(warped)
    x0 = onp.posenc(np.where(valid, warped, 0.0) / hp[0], int(hp[4]))
    x, ties = x0, 0
    points = np.zeros(valid.shape[:-1], dtype=bool)
    skip_layer = depth // 2
    for i in range(depth):
        k, b = tree['Dense_%d' % i]['kernel'].astype(np.float64), tree['Dense_%d' % i]['bias'].astype(np.float64)
        a = x @ k + b
        mag = np.abs(x) @ np.abs(k) + np.abs(b)
        near = (np.abs(a) <= rel * mag) & valid[..., :1]
        ties += int(near.sum())
        points |= near.any(axis=-1)
        x = np.maximum(a, 0.0)
        if i % skip_layer == 0 and i > 0:
            x = np.concatenate([x, x0], axis=-1)
    return (ties, points.any(axis=0)) if return_points else ties
