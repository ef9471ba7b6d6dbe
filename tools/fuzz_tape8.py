"""Randomised sweep of the 8-bit tape mode (BHN_BF16_T8; not part of the test suite): N random 256-wide problems of
tests/test_gpu_backward.py::random_problem (depth 4 / 6 / 8, 0-3 Stokes planes, posenc degree 0-4, random ray grids, samples per
ray, frames and recovery domains).  Per draw: the mode's gradient must be finite, inside the bf16 mode's bounds against the
float64 oracle (x2: those bounds are statistical on problems this small), and is compared with the bf16 mode's own gradient --
the rounding of the 8-bit operands (~3 % per element) averages out over the points of a problem, so the difference is printed
against the number of points inside the recovery domain.
    python tools/fuzz_tape8.py [N] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_backward as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
bad = 0
rows = []
for i in range(N):
    depth, S, deg = int(rng.choice([4, 6, 8])), int(rng.integers(0, 4)), int(rng.integers(0, 5))
    T.RANDOM_PROBLEM_SHAPE = (int(rng.integers(3, 12)), int(rng.integers(3, 10)), int(rng.choice([int(rng.integers(3, 141)), 32, 33, 64, 65, 100])),
                              int(rng.integers(1, 5)))
    T.RANDOM_PROBLEM_JITTER = tuple(rng.uniform(-0.05, 0.05, 3))
    T.RANDOM_PROBLEM_DOMAIN = (float(rng.uniform(4.0, 12.0)), float(rng.choice([0.0, rng.uniform(0.5, 5.0)])),
                               float(rng.choice([np.inf, rng.uniform(5.5, 14.0)])), float(rng.choice([np.inf, rng.uniform(1.0, 8.0)])))
    prob = T.random_problem(256, depth, S, deg)
    if float(np.abs(prob['gref']).max()) == 0.0:
        continue                                              # empty recovery domain
    g16, g8 = [], []
    try:
        T.random_problem_errors(prob, 'bf16', dev, g16)
        i8, gerr, l2 = T.random_problem_errors(prob, 'bf16_t8', dev, g8)
    except Exception as e:                                    # noqa: BLE001
        bad += 1
        print('HARD draw %d depth %d S %d deg %d shape %s: %s' % (i, depth, S, deg, T.RANDOM_PROBLEM_SHAPE, str(e).split('\n')[0][:160]), flush=True)
        continue
    d = T.l2err(g8[0], g16[0])
    H, W, G, B = T.RANDOM_PROBLEM_SHAPE
    ok = np.isfinite(g8[0]).all() and gerr < 2 * T.GTOL['bf16'] and l2 < 2 * T.L2TOL['bf16'] and d < 0.15
    rows.append((H * W * G * B, d, l2))
    if not ok:
        bad += 1
        print('HARD draw %d depth %d S %d deg %d shape %s domain %s: vs oracle max %.2e L2 %.2e, vs bf16 %.2e' %
              (i, depth, S, deg, T.RANDOM_PROBLEM_SHAPE, tuple(round(v, 2) for v in T.RANDOM_PROBLEM_DOMAIN), gerr, l2, d), flush=True)
rows.sort()
print('%d draws with a non-empty domain; difference to the bf16 gradient (relative L2) by problem size:' % len(rows))
for lo, hi in ((0, 1000), (1000, 3000), (3000, 10000), (10000, 10 ** 9)):
    sel = [r for r in rows if lo <= r[0] < hi]
    if sel:
        print('  %6d .. %-10s ray samples: %3d draws, median %.2e, worst %.2e (vs oracle: worst L2 %.2e)' %
              (lo, hi if hi < 10 ** 9 else '', len(sel), float(np.median([r[1] for r in sel])), max(r[1] for r in sel), max(r[2] for r in sel)))
print('%d of %d draws failed' % (bad, N))
sys.exit(1 if bad else 0)
