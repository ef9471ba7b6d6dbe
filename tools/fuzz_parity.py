"""Randomised parity sweep (not part of the test suite): tests/test_gpu_backward.py::test_random_problem_f32_and_bf16 --
images and gradients of both arithmetic modes against the float64 oracle on a ragged problem -- for N random
(width 1..256, depth 2..8, Stokes planes 0..3, posenc degree 0..4) combinations.  The bf16
bounds of the test are statistical on this tiny problem (3 x 63 rays): a few per cent of the draws exceed them by
less than a factor of two; anything beyond that, and any f32 failure, is reported as HARD (exit code 1).  f32 draws with
detected ReLU ties are adjudicated by the test itself (tests/test_gpu_backward.py::adjudicate_relu_ties) and print a
"relu ties adjudicated" line.
    python tools/fuzz_parity.py [N] [seed]          (FUZZ_ONLY=i,j,...: run only these draws of the sequence)
FUZZ_GENERAL=1: every draw is a shape OUTSIDE the fused kernels (posenc degree 5..10 and / or width 257..512: csrc/general_mlp.hip),
held to tests/test_gpu_backward.py::test_shapes_outside_the_fused_kernels (f32 mode: f32 bounds; bf16 mode -- its own MFMA kernels since
round 6 -- the bf16 bounds; each mode bitwise reproducible run to run)."""
import os, sys, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_backward as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
bad = soft = 0
GENERAL = bool(os.environ.get('FUZZ_GENERAL'))
ONLY = set(int(v) for v in os.environ.get('FUZZ_ONLY', '').split(',') if v)        # run only these draws (the RNG sequence is replayed for all)
for i in range(N):
    width = int(rng.choice([int(rng.integers(1, 257)), 32, 64, 128, 256]))
    depth, S, deg = int(rng.integers(2, 9)), int(rng.integers(0, 4)), int(rng.integers(0, 5))
    if GENERAL:
        kind = int(rng.integers(0, 3))                      # 0: high degree, 1: wide, 2: both
        if kind != 1: deg = int(rng.integers(5, 11))
        if kind != 0: width = int(rng.integers(257, 513))
    # ray grid, samples per ray (1 .. 140: below / at / above the 32- and 64-sample tile boundaries), frames
    T.RANDOM_PROBLEM_SHAPE = (int(rng.integers(2, 12)), int(rng.integers(2, 10)), int(rng.choice([int(rng.integers(3, 141)), 32, 33, 64, 65, 100])),
                              int(rng.integers(1, 5)))
    T.RANDOM_PROBLEM_JITTER = tuple(rng.uniform(-0.05, 0.05, 3))       # no sample exactly on a domain boundary (mask ties)
    # recovery domain: scale, rmin, rmax, z_width -- from "everything inside" (no compaction) to a thin shell
    T.RANDOM_PROBLEM_DOMAIN = (float(rng.uniform(4.0, 12.0)), float(rng.choice([0.0, rng.uniform(0.5, 5.0)])),
                               float(rng.choice([np.inf, rng.uniform(5.5, 14.0)])), float(rng.choice([np.inf, rng.uniform(1.0, 8.0)])))
    T.RANDOM_PROBLEM_SHAPE = (max(T.RANDOM_PROBLEM_SHAPE[0], 3), max(T.RANDOM_PROBLEM_SHAPE[1], 3)) + T.RANDOM_PROBLEM_SHAPE[2:]
    # (two rows / columns sit at alpha or beta = +-8: nothing inside the domain, all-zero images)
    if ONLY and i not in ONLY:
        continue
    try:
        (T.test_shapes_outside_the_fused_kernels if GENERAL else T.test_random_problem_f32_and_bf16)(dev, width, depth, S, deg)
        if ONLY:
            print('draw %d width %d depth %d S %d deg %d shape %s: passed' % (i, width, depth, S, deg, T.RANDOM_PROBLEM_SHAPE), flush=True)
    except Exception as e:                                   # noqa: BLE001 -- report and go on
        msg = str(e).split('\n')[0][:200]
        hard = True
        try:
            mode, err = e.args[0][0], float(e.args[0][1])
            # every f32 failure is HARD: detected ReLU ties are adjudicated inside the test (weights nudged until the float64
            # forward has no tie; the f32 gradient must then meet 2e-5) and only un-explained ones get here
            hard = mode == 'f32' or (mode == 'bf16' and err > 2 * T.GTOL['bf16'])
        except Exception:                                    # noqa: BLE001
            pass
        soft += not hard
        bad += hard
        print('%s draw %d width %d depth %d S %d deg %d shape %s domain %s: %s' % ('HARD' if hard else 'soft', i, width, depth, S, deg, T.RANDOM_PROBLEM_SHAPE, tuple(round(v, 2) for v in T.RANDOM_PROBLEM_DOMAIN), msg), flush=True)
print('%d of %d random configurations failed hard, %d exceeded a bf16 bound by less than 2x' % (bad, N, soft))
sys.exit(1 if bad else 0)
