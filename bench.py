#!/usr/bin/env python3
"""Benchmark of the hot path: train ray-samples/sec of the image-plane recovery step.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process only spawns `python -m torch.distributed.run
--nproc-per-node N bench.py ...` (before it touches a GPU itself) and relays rank 0's JSON line; launched by
torch.distributed.run (the driver's form) the ranks run directly.  Either way WORLD_SIZE must equal --gpus.

Workload (BASELINE.json configs[1], "Tutorial3 image-plane recovery"): 128x128 rays x 64 samples per
ray, 64 frames, 4x256 MLP, bf16 MFMA, 8 frames per GPU per step (weak scaling: the frame batch is
8*N, the reference's own time-frame data parallelism), loss 'full', Adam with linear decay.
One step = TemporalBatchedArgs.sample -> pack weights -> training forward (records the tape) -> chi^2 -> delta chain ->
dW GEMM -> slab reduce -> RCCL all-reduce of the flat gradient (N>1) -> Adam.
Inputs are synthetic geodesic arrays resident in HBM before the timed region; the
"all-active" variant (rmin=0, rmax=inf, z_width=inf, nothing pre-injection) is used so that every
ray-sample goes through the full MLP (evaluated points == total points).

Prints ONE JSON line on rank 0.  `roofline` refers to the MLP kernel with the largest share of the step.  SURVEY 8(d) puts the
fused MLP, forward and backward, on the MFMA roofline: `roofline.bound` is "mfma", `achieved` = ALGORITHMIC flops per launch
(SURVEY 8(d)'s per-point figure x the points one launch evaluates) / that kernel's average launch duration, `frac` =
achieved / the dense bf16 (or f32) MFMA peak.  The tape bytes this design chose to move are NOT algorithmic work: they are
reported beside it as `roofline.tape_stream` (GB/s of tape, its fraction of 8 TB/s, bytes per point) and never as `frac`.
Every kernel is timed live with HIP events on the launch stream (`bhn_render_bwd_tape_timed` records the caller's events
between the kernels of the backward); `roofline.step_mfma_frac` is the whole step on the same basis.  Keys ending in
`_from_profiles` (and `traffic`) are NOT measured in this run: they are read from the committed rocprofv3 counter passes
`profiles/<roofline.profiles_tag>_*` (PMC counters cannot be collected from inside the process) and are flagged
`profiles_match_this_build: false` when the library they were collected on is not the one loaded here.
`fwd_images_per_s` is frames / time of `optimization.total_movie_loss` (the reference's test path, optimization.py:14-66)
over the whole movie; `parity_mode` is the same step in the f32 (1e-5 parity) arithmetic; `tape8_mode` the same step with the
backward's tape in 8 bits (BHN_BF16_T8: bf16 arithmetic, e4m3 dW operands -- an A/B beside the headline, never the headline);
`cpu_baseline` is the oracle's PyTorch-CPU restatement timed on the host cores on a bounded sample BEFORE the GPU work starts
(best of a thread-count sweep): `value` for the training step, `fwd_images_per_s` for the forward (test) path.
`roofline.kernels[*]`: each MLP kernel timed INSIDE the K timed steps (HIP events on the launch stream, engine.step_timer), the
flops it executes of the algorithm (kernel_flops), the tape bytes the library's layout makes it stream (bhn_tape_info) and the
core clock it sustained (`sustained_clock_mhz`: s_memtime / s_memrealtime stamps of its workgroup 0, bhn_frames.clock_probe);
`roofline.mfma_peak_this_box`: a register-operand bf16 MFMA loop on random data run in the same process (bhn_mfma_probe) -- what
the matrix pipes of THIS board deliver under its power cap, against the 2.5 PFLOP/s the fractions are quoted on.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from kernel_names import short_name, FWD_NAME, CHAIN_NAME, FUSED_NAME, INFER_NAME      # (kernel names as rocprofv3 prints them)

PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
PROFILE_TAG = 'r6'                                  # profiles/<tag>_pmc_traffic.json etc. (tools/collect_profiles.sh)


def mlp_flops(depth, width, F=21):
    """Algorithmic MLP flops per evaluated point: forward, delta chain, weight gradient, training step."""
    fwd = 2 * (F * width + (depth - 2) * width * width + (width + F) * width + width)
    chain = 2 * (depth - 1) * width * width
    train = 3 * fwd - 2 * F * width
    return fwd, chain, train - fwd - chain, train


def kernel_flops(depth, width, flags, F=21):
    """Flops per evaluated point that each MLP kernel of the step EXECUTES of the algorithm (no recompute, no padding): the
    forward; the delta chain = the hidden layers' transposed products + the output layer's delta (2 W) -- and, where the chain
    accumulates dW_0 itself (TapeLayout::ga0_chain), layer 0's weight gradient (2 F W); the weight-gradient kernel = every
    layer's dW (= the forward's flops) minus what the chain took over.  SURVEY 8(d)'s step figure, 3 fwd - 2 F W, also counts
    an input gradient of the skip layer's encoded features (2 F W) that no kernel needs: the per-kernel figures add up to
    2 F W less than it.  The fused width-128 backward executes chain + dW in one kernel."""
    fwd = mlp_flops(depth, width, F)[0]
    chain = 2 * (depth - 1) * width * width + 2 * width
    dw = fwd
    if flags.get('ga0_chain'):
        chain += 2 * F * width
        dw -= 2 * F * width
    if flags.get('fused128'):
        return {FWD_NAME: fwd, FUSED_NAME: chain + dw}
    return {FWD_NAME: fwd, CHAIN_NAME: chain, 'dw_kernel': dw}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--mode', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--image', type=int, default=128)
    ap.add_argument('--ngeo', type=int, default=64)
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--frames-per-gpu', type=int, default=8)
    ap.add_argument('--width', type=int, default=256)
    ap.add_argument('--depth', type=int, default=4)
    ap.add_argument('--masked', action='store_true', help='use the tutorial domain masks instead of all-active')
    ap.add_argument('--overlap-allreduce', action='store_true',
                    help='N > 1: run the all-reduce of step k under step k+1 (one-step-stale gradients; NOT the reference semantics, off by default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-mode', action='store_true', help='skip the f32 parity-mode block')
    ap.add_argument('--no-tape8', action='store_true', help='skip the 8-bit-tape A/B block')
    ap.add_argument('--no-width128', action='store_true', help='skip the block with the same step on the reference-default 4x128 network')
    ap.add_argument('--no-tutorial-domain', action='store_true',
                    help='skip the masked-domain variant (profiling runs: keeps every launch of a kernel the same shape)')
    ap.add_argument('--no-other-configs', dest='other_configs', action='store_false',
                    help='skip the block with one timed training step each at the sizes of BASELINE configs 3 (256x256x128, Stokes '
                         'I/Q/U, lc), 4 (EHT2017 visibilities, 256x256x100) and 5 (64x64x100, 4x128, lc) on this one GPU (N = 1 only)')
    ap.add_argument('--other-configs', dest='other_configs', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-rays', type=int, default=2048, help='rays of one frame in the CPU-baseline sample')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='time budget of the CPU-baseline thread sweep')
    return ap.parse_args()


def spawn_ranks(args):
    """`bench.py --gpus N` from a plain shell: start N ranks with torch.distributed.run and relay rank 0's line.
    Runs before this process has made a HIP call of its own that creates a context, and never replaces itself: the ranks
    are a fresh child process tree (device_count() may fall back to hipGetDeviceCount on this image; harmless here)."""
    import socket
    n = args.gpus
    one_dev = os.environ.get('BHNERF_BENCH_ONE_DEVICE') == '1'
    ndev = torch.cuda.device_count()
    if ndev < n and not one_dev:
        raise SystemExit('bench.py --gpus %d: only %d device(s) visible (BHNERF_BENCH_ONE_DEVICE=1 runs all ranks on '
                         'cuda:0 over gloo, a testing aid)' % (n, ndev))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    if res.returncode != 0 or not lines:
        sys.stderr.write(res.stdout)
        return res.returncode or 1
    print(lines[-1])
    return 0


def cpu_baseline(args, geo, GM_c3):
    """The oracle's torch-CPU restatement (5 un-fused GEMMs + autograd + Adam, float32: the analogue of the reference's
    XLA-CPU execution) on a bounded sample of the workload.  The points are one flat (rays*samples, features) matrix, so
    every layer is ONE 2-D GEMM.  Thread counts {8,16,32,64,128,all} are tried (more threads than the GEMMs can feed make
    the step slower: 256 threads measured 7.7x slower than 8 in round 1); the best is reported with its count."""
    from oracle import oracle_np as onp
    from oracle import oracle_torch as ot
    try:
        ncores = len(os.sched_getaffinity(0))      # cores this process may actually use
    except AttributeError:
        ncores = os.cpu_count() or 1
    R = args.image * args.image
    nr = min(args.cpu_rays, R)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), dtype=torch.float32)
    sub = lambda v: t(v.reshape((-1,) + v.shape[-1:])[:nr])                          # (rays, samples)
    geom = dict(coords=torch.stack([sub(geo['coords'][i]) for i in range(3)]), Omega=sub(geo['Omega']),
                t_geos=sub(geo['t_geos']), g=sub(geo['g']), dtau=sub(geo['dtau']), Sigma=sub(geo['Sigma']), J=None,
                t_start_obs=0.0, t_injection=float(geo['t_injection']))
    hp = dict(GM_c3=GM_c3, scale=args.fov / 2, rmin=0.0, rmax=float('inf'), z_width=float('inf'), posenc_deg=3,
              net_depth=args.depth)
    rng = np.random.default_rng(1)
    tree = onp.he_uniform_params(rng, args.depth, args.width, 21)
    tf = torch.tensor([0.3], dtype=torch.float32)
    target = torch.zeros((1, nr)); sigma = torch.ones((1, nr))
    default_threads = torch.get_num_threads()
    sweep, t_start = {}, time.perf_counter()
    for nthr in sorted({n for n in (8, 16, 32, 64, 128, ncores) if n <= ncores} or {ncores}):
        tried = sorted(sweep)
        if sweep and (time.perf_counter() - t_start > args.cpu_seconds or
                      (len(tried) >= 3 and all(sweep[n] > 1.05 * min(sweep.values()) for n in tried[-2:]))):
            break                              # out of budget, or the last TWO (larger) counts were slower than the best
                                               # (one is not enough: the curve is not monotonic -- 16 threads slower than 8, 32 faster)
        torch.set_num_threads(nthr)
        ks, bs = ot.tree_to_lists(tree, torch.float32)
        tr = ot.CpuTrainer(ks, bs, geom, hp, num_iters=1000)
        tr.step(tf, target, sigma, target, 1.0, 'full')                      # warm-up
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            tr.step(tf, target, sigma, target, 1.0, 'full')
            times.append(time.perf_counter() - t0)
        sweep[nthr] = min(times)
    best = min(sweep, key=sweep.get)
    dt = sweep[best]
    # the second half of BASELINE's metric, `fwd images/sec`: the test path (optimization.py:14-66: forward render + chi^2 per
    # batch of frames, no gradient) on the same sample at the best thread count, scaled to whole frames of R rays
    torch.set_num_threads(best)
    ks, bs = ot.tree_to_lists(tree, torch.float32)
    tr = ot.CpuTrainer(ks, bs, geom, hp, num_iters=1000)
    fwd_times = []
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            img = tr.forward(tf)
            ot.loss_image(img, target, sigma, target, 1.0, 'full')
            fwd_times.append(time.perf_counter() - t0)
    dt_f = min(fwd_times[1:])
    torch.set_num_threads(default_threads)
    return {'value': nr * args.ngeo / dt, 'unit': 'ray-samples/s', 'cores': best, 'host_cores': ncores, 'kind': 'port',
            'fwd_images_per_s': round(nr / (R * dt_f), 4),
            'fwd_sample': 'forward render + chi^2 (no gradient) of the same %d-ray sample at %d threads: %.3f s, scaled to frames of %d rays '
                          '(the path of optimization.total_movie_loss, optimization.py:14-66)' % (nr, best, dt_f, R),
            'threads_sweep_s_per_step': {str(k): round(v, 4) for k, v in sweep.items()},
            'sample': '1 frame x %d rays x %d samples as one (%d, features) matrix, 4x%d MLP, float32 torch-CPU fwd+bwd+Adam, '
                      'best of 2 steps at the best of the thread counts tried (%.3f s/step at %d threads)'
                      % (nr, args.ngeo, nr * args.ngeo, args.width, dt, best)}


def file_md5(path):
    import hashlib
    try:
        return hashlib.md5(open(path, 'rb').read()).hexdigest()
    except OSError:
        return None


def _hip_lib_path():
    from bhnerf_amd import _hip
    return os.environ.get('BHNERF_HIP_LIB') or _hip.LIB_PATH


def other_configs(dev, mode):
    """One timed training step (mean of 3 after 2 warm-up steps) at the sizes of BASELINE configs 3, 4 and 5 on this one
    GPU, through the reference-shaped API (TrainStep / Optimizer): the multi-GPU configs shard frames, so the per-GPU
    share of a step is 8 frames here as well.  Synthetic geodesics, tutorial-style recovery domains."""
    from bhnerf_amd import constants, network, observation, optimization, synthetic, units
    GM = constants.GM_c3('hr')
    B = 8
    out = {}

    def time_steps(step, opt, rt, n_warm=2, n=3, nb=B):
        idx = np.arange(nb)
        for _ in range(n_warm):
            opt.loss, opt.state, _ = step(opt.state, rt, idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            opt.loss, opt.state, _ = step(opt.state, rt, idx)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    lc_cfgs = {'config3': dict(H=256, W=256, G=128, width=256, fov=40.0, inc=60.0, spin=0.94, rmin=2.024, rmax=20.0, z_width=4.0),
               'config5': dict(H=64, W=64, G=100, width=128, fov=40.0, inc=12.0, spin=0.0, rmin=6.0, rmax=20.0, z_width=4.0)}
    for name, c in lc_cfgs.items():
        geo = synthetic.synthetic_geodesics(c['H'], c['W'], c['G'], fov_M=c['fov'], inc_deg=c['inc'], spin=c['spin'], S=3, seed=3)
        t_frames = np.linspace(0.0, 1.7, 128)[:B]
        pred = network.NeRF_Predictor(c['rmax'], c['rmin'], c['rmax'], c['z_width'], net_depth=4, net_width=c['width'], mode=mode, device=dev)
        rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                          t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=geo['J'])
        target = np.random.default_rng(5).uniform(0.5, 1.5, (B, 3)).astype(np.float32)
        step = optimization.TrainStep.image(t_frames * units.hr, target, sigma=0.1, dtype='lc')
        opt = optimization.Optimizer({'num_iters': 100, 'lr_init': 1e-4, 'lr_final': 1e-6}, pred, rt)
        small = c['H'] <= 64
        dt = time_steps(step, opt, rt, n=20 if small else 3)
        gm = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], rt['J'], rt['g'], rt['dtau'], rt['Sigma'])
        n = B * c['H'] * c['W'] * c['G']
        out[name] = {'workload': '%dx%d rays x %d samples, Stokes I/Q/U, loss lc, 4x%d MLP, %d frames/step' % (c['H'], c['W'], c['G'], c['width'], B),
                     'dtype': mode, 'ms_per_step': round(1e3 * dt, 3), 'value': round(n / dt, 1), 'unit': 'ray-samples/s',
                     'active_fraction': round(gm.active_fraction, 4), 'visited_fraction': round(gm.visited_fraction, 4),
                     'tape_frame_group': (B if pred.engine().fits_tape(B, gm.P_eff) else pred.engine().tape_group(B, gm.P_eff))}
        # the same steps captured into a HIP graph (hparams['hip_graph']): eager and graph side by side
        step.use_graph = True
        try:
            dtg = time_steps(step, opt, rt, n_warm=3, n=20 if small else 3)
            if any(step._graphs.values()):
                out[name]['ms_per_step_hip_graph'] = round(1e3 * dtg, 3)
            if small:       # the reference's own batch size (scripts/Fit_ALMA_LP_Apr11_SgrA_Flare.yaml:46: batchsize 6)
                step.use_graph = False
                dt6 = time_steps(step, opt, rt, n=20, nb=6)
                step.use_graph = True
                dt6g = time_steps(step, opt, rt, n_warm=3, n=20, nb=6)
                out[name + '_batch6'] = {'workload': out[name]['workload'].replace('%d frames/step' % B, '6 frames/step'), 'dtype': mode,
                                         'ms_per_step': round(1e3 * dt6, 3), 'ms_per_step_hip_graph': round(1e3 * dt6g, 3),
                                         'value': round(6 * c['H'] * c['W'] * c['G'] / dt6, 1), 'unit': 'ray-samples/s'}
        except Exception as exc:
            out[name]['hip_graph_error'] = repr(exc)
        finally:
            step.use_graph = False
            step.clear_graphs()
        del opt, pred, gm, rt
        torch.cuda.empty_cache()
    # config 4: EHT2017 (u, v) tracks from the reference's station file (fixture g11: data, made by tests/golden/make_eht2017.py)
    NPIX, G4, FOV = 256, 100, 16.0
    g11 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g11_eht2017.npz'))
    frames = np.arange(0, 64, 64 // B)
    t_hr = g11['t_hr'][frames]
    geo = synthetic.synthetic_geodesics(NPIX, NPIX, G4, fov_M=FOV, inc_deg=60.0, seed=0)
    rad_per_M = 5.03e-6 / 3600.0 * np.pi / 180.0
    A = np.stack([observation.dft_matrix(g11['uv'][f], FOV * rad_per_M, NPIX) for f in frames])
    rng = np.random.default_rng(4)
    target = (rng.normal(size=A.shape[:2]) + 1j * rng.normal(size=A.shape[:2])).astype(np.complex64)
    sigma = g11['sigma'][frames].astype(np.float32)
    rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                      t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr)
    pred = network.NeRF_Predictor(FOV / 2, 2.0, FOV / 2, 4.0, net_depth=4, net_width=256, mode=mode, device=dev)
    step = optimization.TrainStep.eht_arrays(t_hr * units.hr, target, sigma, A, dtype='vis')
    opt = optimization.Optimizer({'num_iters': 100, 'lr_init': 1e-4, 'lr_final': 1e-6}, pred, rt)
    dt = time_steps(step, opt, rt)
    gm = pred.geometry(rt['coords'], rt['Omega'], rt['t_geos'], None, rt['g'], rt['dtau'], rt['Sigma'])
    n = B * NPIX * NPIX * G4
    out['config4'] = {'workload': "EHT2017 complex-visibility loss ('vis', 28 baselines, direct-DFT A), %dx%d rays x %d samples, 4x256 MLP, %d frames/step" % (NPIX, NPIX, G4, B),
                      'dtype': mode, 'ms_per_step': round(1e3 * dt, 3), 'value': round(n / dt, 1), 'unit': 'ray-samples/s',
                      'active_fraction': round(gm.active_fraction, 4), 'visited_fraction': round(gm.visited_fraction, 4)}
    return out


def strong_scaling_share(dev, mode='bf16'):
    """The per-GPU share of the reference's own 8-device runs -- b = 8 frames split over 8 devices = ONE frame per GPU and step
    (optimization.py:289-291, 360-362) -- at the shapes of BASELINE configs 3 and 5, on this one GPU (the all-reduce of the
    gradient is not in these numbers).  Per shape: wall ms per step of the eager Python driver, wall ms per step with the
    step captured into a HIP graph (hparams['hip_graph'], optimization.GraphedImageStep), and the GPU time of one graph
    replay (HIP events around back-to-back replays: the kernels of the step and the gaps between them)."""
    from bhnerf_amd import network, optimization, synthetic, units
    out = {}
    cfgs = {'config3': dict(H=256, W=256, G=128, width=256, fov=40.0, inc=60.0, spin=0.94, rmin=2.024, rmax=20.0, z_width=4.0),
            'config5': dict(H=64, W=64, G=100, width=128, fov=40.0, inc=12.0, spin=0.0, rmin=6.0, rmax=20.0, z_width=4.0)}
    for name, c in cfgs.items():
        geo = synthetic.synthetic_geodesics(c['H'], c['W'], c['G'], fov_M=c['fov'], inc_deg=c['inc'], spin=c['spin'], S=3, seed=3)
        nt = 128
        t_frames = np.linspace(0.0, 1.7, nt)
        rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                          t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=geo['J'])
        target = np.random.default_rng(5).uniform(0.5, 1.5, (nt, 3)).astype(np.float32)
        res = {'workload': '%dx%d rays x %d samples, Stokes I/Q/U, loss lc, 4x%d MLP, 1 frame per GPU and step' % (c['H'], c['W'], c['G'], c['width']),
               'dtype': mode}
        for graph in (False, True):
            pred = network.NeRF_Predictor(c['rmax'], c['rmin'], c['rmax'], c['z_width'], net_depth=4, net_width=c['width'], mode=mode, device=dev)
            step = optimization.TrainStep.image(t_frames * units.hr, target, sigma=0.1, dtype='lc')
            step.use_graph = graph
            opt = optimization.Optimizer({'num_iters': 1000, 'lr_init': 1e-4, 'lr_final': 1e-6}, pred, rt)
            frames = step.args[0]
            n = 200 if c['H'] <= 64 else 40
            for _ in range(5):
                opt.loss, opt.state, _ = step(opt.state, rt, frames.sample(1))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                opt.loss, opt.state, _ = step(opt.state, rt, frames.sample(1))
            torch.cuda.synchronize()
            res['wall_ms_per_step_' + ('hip_graph' if graph else 'eager')] = round(1e3 * (time.perf_counter() - t0) / n, 4)
            if graph:
                g = [v for v in step._graphs.values() if v]
                if g:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        g[0].graph.replay()
                    e1.record(); torch.cuda.synchronize()
                    res['gpu_ms_per_graph_replay'] = round(e0.elapsed_time(e1) / n, 4)
                    gm = g[0].geom
                    res['active_fraction'] = round(gm.active_fraction, 4)
                    res['process_group'] = GraphedBackend() if g[0].dist else None
                    res['collective_in_graph'] = bool(g[0].collective_in_graph)    # RCCL all-reduce + Adam captured inside the graph
            del opt, pred, step
            torch.cuda.empty_cache()
        if 'gpu_ms_per_graph_replay' in res:
            res['wall_over_gpu_hip_graph'] = round(res['wall_ms_per_step_hip_graph'] / res['gpu_ms_per_graph_replay'], 3)
            res['wall_over_gpu_eager'] = round(res['wall_ms_per_step_eager'] / res['gpu_ms_per_graph_replay'], 3)
        res['ray_samples_per_s_hip_graph'] = round(c['H'] * c['W'] * c['G'] / (res['wall_ms_per_step_hip_graph'] * 1e-3), 1)
        out[name] = res
    return out


def GraphedBackend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_initialized() else None


class HipEvents:
    """Raw HIP events (ctypes on the runtime the process already has loaded) for bhn_render_bwd_tape_timed."""

    def __init__(self, n):
        self.hip = C.CDLL('libamdhip64.so')
        self.ev = (C.c_void_p * n)()
        for i in range(n):
            e = C.c_void_p()
            assert self.hip.hipEventCreate(C.byref(e)) == 0
            self.ev[i] = e

    def elapsed(self, i, j):
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), C.c_void_p(self.ev[i]), C.c_void_p(self.ev[j])) == 0
        return float(ms.value)


class StepTimer:
    """engine.step_timer: times the MLP kernels of every render_train / render_bwd_tape call it sees -- torch events around the
    training forward (one kernel), the library's own event marks between the kernels of the backward (bhn_render_bwd_tape_timed)
    -- so that the per-kernel figures of the bench line come from the SAME steps as `ms_per_step`."""

    def __init__(self, eng):
        from bhnerf_amd import _hip
        self.names = [_hip.lib().bhn_render_bwd_tape_kernel_name_for(C.byref(eng.model), eng.mode, i).decode() for i in range(3)]
        self.reset()

    def reset(self):
        self.fwd, self.bwd, self._open = [], [], None

    def fwd_mark(self, i):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if i == 0:
            self._open = ev
        else:
            self.fwd.append((self._open, ev))

    def bwd_events(self):
        self.bwd.append(HipEvents(4))
        return self.bwd[-1].ev, 4

    def kernel_ms(self, steps):
        """Mean ms per STEP of each kernel (a step may be several calls: frame groups); call after a device synchronisation."""
        acc = {FWD_NAME: sum(a.elapsed_time(b) for a, b in self.fwd) / steps}
        for i, n in enumerate(self.names):
            acc[n] = sum(e.elapsed(i, i + 1) for e in self.bwd) / steps
        acc.pop('-', None)                          # (an empty kernel slot of the fused width-128 backward)
        return acc


def kernel_times(eng, geom, tM0, dimg, reps=5):
    """Per-kernel ms of one rank's step share outside a step loop (the side blocks: parity mode, 8-bit tape): the training
    forward and the kernels of the tape backward, back to back, through the same StepTimer as the headline."""
    B = int(tM0.numel())
    taped = eng.fits_tape(B, geom.P_eff)
    group = B if taped else eng.tape_group(B, geom.P_eff)
    assert group, 'the tape of one frame does not fit the workspace'
    slices = [slice(b0, min(b0 + group, B)) for b0 in range(0, B, group)]
    timer = StepTimer(eng)
    eng.step_timer = timer
    try:
        for rep in range(-1, reps):                       # rep -1: warm-up
            if rep == 0:
                torch.cuda.synchronize(); timer.reset()
            for sl in slices:
                eng.render_train(geom, tM0[sl])
                eng.render_bwd_tape(geom, tM0[sl], dimg[sl].contiguous())
        torch.cuda.synchronize()
    finally:
        eng.step_timer = None
    return timer.kernel_ms(reps), group


def tape_bytes_per_point(info):
    """Tape bytes per evaluated point that each MLP kernel of the step moves through HBM, from the library's own layout
    (bhn_tape_info -> engine.tape_info): what the design streams, written once and read once."""
    if info['flags']['fused128']:
        return {FWD_NAME: info['fwd_write'] / 32.0, FUSED_NAME: info['dw_read'] / 32.0}
    return {FWD_NAME: info['fwd_write'] / 32.0, CHAIN_NAME: (info['chain_write'] + info['chain_read']) / 32.0, 'dw_kernel': info['dw_read'] / 32.0}


CLK_SLOT = {FWD_NAME: 1, CHAIN_NAME: 2, FUSED_NAME: 2, 'dw_kernel': 3, INFER_NAME: 0}      # BHN_CLK_* of include/bhnerf_hip.h


def clock_mhz(stamps, slot):
    """Sustained core clock of the kernel in `slot` from its four stamps {s_memtime, s_memrealtime} x {start, end} of workgroup 0
    (s_memrealtime counts at 100 MHz)."""
    t0, r0, t1, r1 = (int(v) for v in stamps[4 * slot:4 * slot + 4])
    return round(100.0 * (t1 - t0) / (r1 - r0), 1) if r1 > r0 and t1 > t0 else None


def mfma_peak_this_box(dev, seconds=0.25):
    """bhn_mfma_probe: dependent bf16 MFMA chains with every operand in registers, random data, 8 waves on every CU, run for
    `seconds` back to back: the matrix-pipe ceiling of THIS board under its power cap, and the clock it sustains there."""
    from bhnerf_amd import _hip
    lib = _hip.lib()
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    clk = torch.zeros((2 * ncu,), dtype=torch.int64, device=dev)
    sink = torch.zeros((1024,), dtype=torch.float32, device=dev)
    iters = 20000
    run = lambda: _hip.check(lib.bhn_mfma_probe(ncu, iters, _hip.ptr(clk), _hip.ptr(sink), _hip.stream_ptr(dev)))
    run(); torch.cuda.synchronize()
    ms, t_end = [], time.perf_counter() + seconds
    while time.perf_counter() < t_end or len(ms) < 3:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    c = clk.cpu().numpy().reshape(ncu, 2)
    flop = ncu * 8 * iters * 16 * 32768.0
    return {'tflops': round(flop / (float(np.median(ms[1:])) * 1e-3) / 1e12, 1), 'clock_mhz': round(float(np.median(100.0 * c[:, 0] / c[:, 1])), 1),
            'launches': len(ms), 'ms_per_launch': round(float(np.median(ms[1:])), 3),
            'what': 'bhn_mfma_probe: %d workgroups x 8 waves x %d x 16 dependent v_mfma_f32_32x32x16_bf16, register operands, random data' % (ncu, iters)}


def attach_profiles(roofline, width, std, lib_md5):
    """What the committed counter passes say about this workload (NOT measured in this run; see the module docstring): `traffic`
    (PMC bytes of the dominant kernel) and the `*_from_profiles` keys, from profiles/<PROFILE_TAG>_*; `profiles_match_this_build`
    compares the library they were collected on with the one loaded here.  Also run offline on a stored bench line
    (tools/attach_profiles.py: the line tools/collect_profiles.sh writes is produced BEFORE the passes it would quote)."""
    dom_k = roofline['kernel']
    step_kernels = [k for k in roofline['kernels']]
    sfx = '_w128' if (width == 128 and not std) else ''      # the width-128 passes of tools/collect_profiles.sh
    use = std or sfx == '_w128'
    prof = {'pmc': {}, 'sq': {}, 'stats': {}, 'match': None}
    traffic_file = 'profiles/%s_pmc_traffic%s.json' % (PROFILE_TAG, sfx)
    try:
        j = json.load(open(os.path.join(ROOT, traffic_file)))
        prof['pmc'] = j['kernels'] if use else {}
        prof['match'] = (j.get('lib_md5') == lib_md5) if j.get('lib_md5') else None
    except Exception:
        pass
    try:
        prof['sq'] = json.load(open(os.path.join(ROOT, 'profiles', PROFILE_TAG + sfx + '_sq_summary.json')))['kernels'] if use else {}
    except Exception:
        pass
    try:                                             # rocprofv3 --kernel-trace --stats averages of the same workload
        import csv
        stats_file = os.path.join(ROOT, 'profiles', PROFILE_TAG + ('_w128' if sfx else '_bench') + '_kernel_stats.csv')
        for r in csv.DictReader(open(stats_file)):
            n = short_name(r['Name'])                 # template arguments by position (tools/kernel_names.py)
            if n and use:
                prof['stats'][n] = float(r['AverageNs']) * 1e-6
    except Exception:
        pass
    roofline.update({'traffic': prof['pmc'].get(dom_k, {}).get('hbm_bytes'),
                     'traffic_source': '%s (rocprofv3 --pmc passes, not measured in this run)' % traffic_file,
                     'profiles_tag': PROFILE_TAG, 'profiles_match_this_build': prof['match']})
    if dom_k in prof['stats']:                       # the same fraction on the rocprofv3 average duration of the committed profile
        ms_p = prof['stats'][dom_k]
        roofline['frac_from_profiles'] = round(roofline['frac'] * roofline['kernel_ms'][dom_k] / ms_p, 4)
        roofline['kernel_ms_from_profiles'] = round(ms_p, 4)
    if prof['pmc']:
        if all(k in prof['pmc'] and 'hbm_bytes' in prof['pmc'][k] for k in step_kernels):
            roofline['step_mlp_traffic_from_profiles'] = int(sum(prof['pmc'][k]['hbm_bytes'] for k in step_kernels))
    if prof['sq']:    # matrix-pipe busy fraction of SIMD cycles from the committed SQ counter passes (keys: short kernel names)
        busy = {k: v['mfma_busy_frac'] for k, v in prof['sq'].items() if 'mfma_busy_frac' in v}
        roofline['mfma_busy_frac_from_profiles'] = busy
        conf = {k: v['lds_bank_conflict_frac'] for k, v in prof['sq'].items() if 'lds_bank_conflict_frac' in v}
        roofline['lds_bank_conflict_frac_from_profiles'] = conf
        pms = {k: v.get('ms') for k, v in prof['sq'].items()}
        if all(k in busy and pms.get(k) for k in step_kernels):
            roofline['step_mfma_busy_frac_from_profiles'] = round(sum(busy[k] * pms[k] for k in step_kernels) / sum(pms[k] for k in step_kernels), 3)


def roofline_block(eng, geom, tM0, dimg, depth, width, mode, frames_per_gpu, std, kern_ms=None, clocks=None):
    """Per-kernel timings of one rank's share of a step (`kern_ms`: measured by a StepTimer inside the caller's step loop; None:
    measured here, back to back) and for each MLP kernel the MFMA fraction on the flops it executes of the algorithm
    (kernel_flops) beside the tape bytes it streams (bhn_tape_info).  The kernel with the largest share of the step is `roofline`."""
    from bhnerf_amd import _hip
    in_loop = kern_ms is not None
    group = frames_per_gpu if eng.fits_tape(frames_per_gpu, geom.P_eff) else eng.tape_group(frames_per_gpu, geom.P_eff)
    clk = torch.zeros((4 * _hip.BHN_CLK_SLOTS,), dtype=torch.int64, device=eng.device)
    if kern_ms is None:
        eng.clock_probe = clk
        try:
            kern_ms, group = kernel_times(eng, geom, tM0, dimg)
        finally:
            eng.clock_probe = None
        clocks = clk.cpu().numpy().copy()
    kern_ms = dict(kern_ms)
    fused = FUSED_NAME in kern_ms

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    eng.clock_probe = clk
    try:
        kern_ms[INFER_NAME] = timed(lambda: eng.render(geom, tM0))
    finally:
        eng.clock_probe = None
    infer_clock = clock_mhz(clk.cpu().numpy(), CLK_SLOT[INFER_NAME])
    pts = frames_per_gpu * geom.P * geom.visited_fraction     # points that go through the MLP
    f_fwd = mlp_flops(depth, width)[0]
    tinfo = eng.tape_info((geom.P_eff + 31) // 32)
    alg = kernel_flops(depth, width, tinfo['flags'])
    assert fused == tinfo['flags']['fused128']
    bpp = tape_bytes_per_point(tinfo)
    peak = PEAK_TFLOPS[mode]
    per = {}
    for k in alg:
        tf = alg[k] * pts / (kern_ms[k] * 1e-3) / 1e12
        gb = bpp[k] * pts / (kern_ms[k] * 1e-3) / 1e9
        per[k] = {'ms': round(kern_ms[k], 4), 'flop_per_point': alg[k], 'mfma_tflops': round(tf, 1), 'mfma_frac': round(tf / peak, 4),
                  'tape_GB_per_s': round(gb, 1), 'hbm_frac': round(gb / 8000.0, 4), 'tape_bytes_per_point': round(bpp[k], 1),
                  'closer_to': 'hbm (tape stream)' if gb / 8000.0 > tf / peak else 'mfma',
                  'sustained_clock_mhz': clock_mhz(clocks, CLK_SLOT[k]) if clocks is not None else None}
    dom_k = max(alg, key=lambda k: kern_ms[k])
    d = per[dom_k]
    # the fused MLP is on the MFMA roofline, on ALGORITHMIC flops; the tape bytes are design overhead, shown beside it
    roofline = {'bound': 'mfma', 'kernel': dom_k, 'achieved': d['mfma_tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': d['mfma_frac'],
                'algorithmic_flop_per_point': alg[dom_k], 'sustained_clock_mhz': d['sustained_clock_mhz'],
                'tape_layout': {k: v for k, v in tinfo['flags'].items() if v},
                'tape_stream': {'GB_per_s': d['tape_GB_per_s'], 'frac_of_8TBps': d['hbm_frac'], 'tape_bytes_per_point': d['tape_bytes_per_point'],
                                'note': 'bytes of tape this kernel moves through HBM per evaluated point (DESIGN.md 3): the design\'s own traffic, not algorithmic work'}}
    roofline['points_per_launch'] = int(pts)
    roofline['kernels'] = per
    roofline['kernel_ms'] = {k: round(v, 4) for k, v in kern_ms.items()}
    roofline['kernel_ms_sum'] = round(sum(v for k, v in kern_ms.items() if 'inference' not in k), 4)
    roofline['kernel_ms_note'] = ('HIP events around / between the kernels INSIDE the timed steps (engine.step_timer): the same steps as ms_per_step' if in_loop
                                  else 'each kernel timed in a loop of its own (HIP events around / between the launches)')
    attach_profiles(roofline, width, std, file_md5(_hip_lib_path()))
    inf_tf = f_fwd * pts / (kern_ms[INFER_NAME] * 1e-3) / 1e12
    roofline['inference_forward'] = {'ms': round(kern_ms[INFER_NAME], 4), 'mfma_tflops': round(inf_tf, 1), 'mfma_frac': round(inf_tf / peak, 4),
                                     'sustained_clock_mhz': infer_clock}
    roofline['_std'] = std
    roofline['_alg'] = alg
    return roofline, kern_ms, group



def main():
    args = parse()
    args.fov = 16.0
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus' % (args.gpus, world))

    from bhnerf_amd import constants, synthetic
    H = W = args.image
    G, nt = args.ngeo, args.frames
    geo = synthetic.synthetic_geodesics(H, W, G, fov_M=args.fov, inc_deg=60.0, seed=0)
    GM_c3 = constants.GM_c3('hr')
    # ---- CPU leg first: the host cores are idle (nothing has touched the GPU yet), bounded by --cpu-seconds -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, geo, GM_c3)

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    # BHNERF_BENCH_ONE_DEVICE=1 (testing aid on a 1-GPU box): every rank uses cuda:0 and the gloo backend, so the
    # multi-process path (frame sharding, gradient all-reduce, identical Adam) runs with the real kernels
    one_dev = os.environ.get('BHNERF_BENCH_ONE_DEVICE') == '1'
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    # BHNERF_BENCH_FORCE_DIST=1 with --gpus 1: a process group of ONE rank over RCCL, so that a 1-GPU box runs the shipped
    # multi-GPU path (init_process_group('nccl', device_id=...), the flat all-reduce on the device, then Adam)
    force_dist = world == 1 and os.environ.get('BHNERF_BENCH_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if force_dist and 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        if one_dev:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus
    backend = dist.get_backend() if dist.is_initialized() else None

    from bhnerf_amd import engine, network, optimization, units
    t_frames = np.linspace(0.0, 1.0, nt)
    rmax = args.fov / 2
    dom = (rmax, 2.0, rmax, 4.0) if args.masked else (rmax, 0.0, np.inf, np.inf)
    pred = network.NeRF_Predictor(*dom, net_depth=args.depth, net_width=args.width, mode=args.mode, device=dev)
    rt_args = network.raytracing_args(
        dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
             t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=1.0)
    target = synthetic.hotspot_movie(geo, t_frames[::max(1, nt // 8)], GM_c3)     # a few distinct frames,
    target = np.ascontiguousarray(np.resize(target, (nt, H, W)))                   # tiled over the movie
    train_step = optimization.TrainStep.image(t_frames * units.hr, target, sigma=1.0, dtype='full')
    hparams = {'num_iters': 5000, 'lr_init': 1e-4, 'lr_final': 1e-6, 'seed': 1, 'overlap_allreduce': args.overlap_allreduce}
    opt = optimization.Optimizer(hparams, pred, rt_args)
    batch = args.frames_per_gpu * world
    assert batch <= nt, 'frames per step exceed the movie length'

    def run_steps(o, n):
        for _ in range(n):
            o.loss, o.state, _ = train_step(o.state, rt_args, indices=train_step.args[0].sample(batch))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the MLP kernels are timed INSIDE the K steps (HIP events on the launch stream around / between them: engine.step_timer)
    # and workgroup 0 of each stamps its clock counters (bhn_frames.clock_probe): roofline.frac and ms_per_step are one loop
    from bhnerf_amd import _hip
    eng = pred.engine()
    timer = None if eng.tape_info(1)['flags']['general'] else StepTimer(eng)      # (the general path has no per-kernel event marks)
    clk_main = torch.zeros((4 * _hip.BHN_CLK_SLOTS,), dtype=torch.int64, device=dev)
    eng.step_timer, eng.clock_probe = timer, clk_main
    run_steps(opt, args.warmup)
    opt.state.finish_allreduce()
    barrier()
    if timer:
        timer.reset()
    t0 = time.perf_counter()
    run_steps(opt, args.steps)
    opt.state.finish_allreduce()        # (--overlap-allreduce: the last gradient's all-reduce + Adam belong to the K steps)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.step_timer, eng.clock_probe = None, None
    kern_loop = timer.kernel_ms(args.steps) if timer else None
    clocks_loop = clk_main.cpu().numpy().copy()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_step = 1e3 * elapsed / args.steps
    samples_step = batch * H * W * G
    value = samples_step / (elapsed / args.steps)
    loss_now = float(torch.as_tensor(opt.loss).float().mean())

    # ---- per-kernel timing of one rank's share (HIP events on the launch stream) ----------------
    geom = pred.geometry(rt_args['coords'], rt_args['Omega'], rt_args['t_geos'], None, rt_args['g'], rt_args['dtau'],
                         rt_args['Sigma'])
    tM0 = engine.frame_offsets(t_frames[:args.frames_per_gpu], 0.0, geo['t_injection'], GM_c3, dev)
    dimg = torch.rand((args.frames_per_gpu, 1, geom.R), device=dev) * 1e-3
    eng.pack(opt.state.flat)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    roofline, kern_ms, group = roofline_block(eng, geom, tM0, dimg, args.depth, args.width, args.mode, args.frames_per_gpu,
                                              std=(H == 128 and G == 64 and args.frames_per_gpu == 8 and args.width == 256 and args.depth == 4
                                                   and args.mode == 'bf16' and not args.masked), kern_ms=kern_loop, clocks=clocks_loop)
    if world == 1 and args.mode == 'bf16':
        try:
            roofline['mfma_peak_this_box'] = mfma_peak_this_box(dev)
        except Exception as exc:                     # a side block must never cost the headline line
            roofline['mfma_peak_this_box'] = {'error': repr(exc)}
    std = roofline.pop('_std')
    f_train = mlp_flops(args.depth, args.width)[3]
    pts = args.frames_per_gpu * geom.P * geom.visited_fraction
    alg = roofline.pop('_alg')
    step_tflops = f_train * value * geom.visited_fraction / 1e12 / world
    roofline['step_algorithmic_tflops'] = round(step_tflops, 2)
    roofline['step_mfma_frac'] = round(step_tflops / PEAK_TFLOPS[args.mode], 4)      # 1,234,944 flop/point at 4x256

    # ---- fwd images/sec: frames / time of the reference's test path over the whole movie (optimization.py:14-66) ----
    fwd_path = None
    if world == 1:
      try:
        optimization.total_movie_loss(args.frames_per_gpu, opt.state, train_step, rt_args)        # warm-up
        torch.cuda.synchronize()
        reps_f = 3
        t0 = time.perf_counter()
        for _ in range(reps_f):
            movie_loss = optimization.total_movie_loss(args.frames_per_gpu, opt.state, train_step, rt_args)
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t0) / reps_f
        fwd_path = {'value': round(nt / dt_f, 1), 'unit': 'images/s', 'frames': nt, 'batch': args.frames_per_gpu,
                    'ms_per_movie': round(1e3 * dt_f, 3), 'movie_loss': movie_loss,
                    'path': 'optimization.total_movie_loss -> TrainStep(update_state=False) -> pack, fused render, chi^2 per batch',
                    'kernel_only_images_per_s': round(args.frames_per_gpu / (kern_ms[INFER_NAME] * 1e-3), 1)}
      except Exception as exc:                       # a side block must never cost the headline line
        fwd_path = {'error': repr(exc)}

    # ---- the same step in the f32 parity arithmetic (the mode that meets north_star's 1e-5), single GPU ----------
    parity = None
    if world == 1 and args.mode == 'bf16' and not args.no_parity_mode and not args.masked:
        pred_p = network.NeRF_Predictor(*dom, net_depth=args.depth, net_width=args.width, mode='f32', device=dev)
        opt_p = optimization.Optimizer(hparams, pred_p, rt_args)
        run_steps(opt_p, 1)
        torch.cuda.synchronize()
        n_p = max(2, min(args.steps, 3))
        t0 = time.perf_counter()
        run_steps(opt_p, n_p)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_p
        eng_p = pred_p.engine()
        geom_p = pred_p.geometry(rt_args['coords'], rt_args['Omega'], rt_args['t_geos'], None, rt_args['g'], rt_args['dtau'], rt_args['Sigma'])
        eng_p.pack(opt_p.state.flat)
        kms, grp = kernel_times(eng_p, geom_p, tM0, dimg, reps=2)
        tf32 = f_train * samples_step / dt / 1e12
        parity = {'dtype': 'f32', 'ms_per_step': round(1e3 * dt, 3), 'value': round(samples_step / dt, 1), 'unit': 'ray-samples/s',
                  'steps': n_p, 'step_algorithmic_tflops': round(tf32, 2), 'step_mfma_frac': round(tf32 / PEAK_TFLOPS['f32'], 4),
                  'peak_tflops': PEAK_TFLOPS['f32'], 'tape_frame_group': grp,
                  'kernel_ms': {k: round(v, 3) for k, v in kms.items()},
                  'mfma_frac': {k: round(v * pts / (kms[k] * 1e-3) / 1e12 / PEAK_TFLOPS['f32'], 4)
                                for k, v in kernel_flops(args.depth, args.width, eng_p.tape_info(1)['flags']).items()},
                  'loss': float(torch.as_tensor(opt_p.loss).float().mean())}
        del opt_p, pred_p, eng_p, geom_p
        torch.cuda.empty_cache()

    # ---- A/B: the same step with the backward's tape in 8 bits (BHN_BF16_T8).  bf16
    #      arithmetic, e4m3 dW operands: NOT the headline's precision -- reported beside it, never as `value` ----------
    tape8 = None
    if world == 1 and args.mode == 'bf16' and args.width == 256 and args.depth >= 3 and args.depth % 2 == 0 and not args.no_tape8 and not args.masked:
      try:
        pred_8 = network.NeRF_Predictor(*dom, net_depth=args.depth, net_width=args.width, mode='bf16_t8', device=dev)
        opt_8 = optimization.Optimizer(hparams, pred_8, rt_args)
        run_steps(opt_8, max(2, args.warmup))
        torch.cuda.synchronize()
        n_8 = max(2, min(args.steps, 50))
        t0 = time.perf_counter()
        run_steps(opt_8, n_8)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t0) / n_8
        eng_8 = pred_8.engine()
        geom_8 = pred_8.geometry(rt_args['coords'], rt_args['Omega'], rt_args['t_geos'], None, rt_args['g'], rt_args['dtau'], rt_args['Sigma'])
        # the gradient of both modes on the SAME weights, frames and d(loss)/d(images)
        eng.pack(opt.state.flat); eng_8.pack(opt.state.flat)
        eng.render_train(geom, tM0); g16 = eng.render_bwd_tape(geom, tM0, dimg).clone()
        eng_8.render_train(geom_8, tM0); g8 = eng_8.render_bwd_tape(geom_8, tM0, dimg).clone()      # (scales: last training step's ratios x this |dimg|max)
        kms8, _ = kernel_times(eng_8, geom_8, tM0, dimg)
        tf8 = f_train * samples_step / dt8 / 1e12
        tape8 = {'dtype': 'bf16 arithmetic, e4m3 tape of the dW operands (BHN_BF16_T8)', 'ms_per_step': round(1e3 * dt8, 3),
                 'value': round(samples_step / dt8, 1), 'unit': 'ray-samples/s', 'steps': n_8,
                 'step_mfma_frac': round(tf8 / PEAK_TFLOPS['bf16'], 4), 'vs_bf16_step': round(1e3 * dt8 / ms_step, 4),
                 'kernel_ms': {k: round(v, 3) for k, v in kms8.items()},
                 'gradient_rel_l2_vs_bf16': float((g8 - g16).norm() / g16.norm()),
                 'loss': float(torch.as_tensor(opt_8.loss).float().mean())}
        del opt_8, pred_8, eng_8, geom_8
        torch.cuda.empty_cache()
      except Exception as exc:                       # a side block must never cost the headline line
        tape8 = {'error': repr(exc)}

    # ---- the reference's DEFAULT network, 4x128 (network.py:19-20, every tutorial and fit script): the same config-2 geometry,
    #      frames and loss; bf16; its own kernel path (fused delta chain + dW) ----------
    width128 = None
    if world == 1 and std and not args.no_width128:
      try:
        pred_w = network.NeRF_Predictor(*dom, net_depth=4, net_width=128, mode='bf16', device=dev)
        opt_w = optimization.Optimizer(hparams, pred_w, rt_args)
        eng_w = pred_w.engine()
        timer_w, clk_w = StepTimer(eng_w), torch.zeros((4 * _hip.BHN_CLK_SLOTS,), dtype=torch.int64, device=dev)
        eng_w.step_timer, eng_w.clock_probe = timer_w, clk_w
        run_steps(opt_w, max(args.warmup, 3))
        torch.cuda.synchronize()
        timer_w.reset()
        n_w = max(10, args.steps // 2)
        t0 = time.perf_counter()
        run_steps(opt_w, n_w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_w
        eng_w.step_timer, eng_w.clock_probe = None, None
        geom_w = pred_w.geometry(rt_args['coords'], rt_args['Omega'], rt_args['t_geos'], None, rt_args['g'], rt_args['dtau'], rt_args['Sigma'])
        eng_w.pack(opt_w.state.flat)
        roof_w, _, grp_w = roofline_block(eng_w, geom_w, tM0, dimg, 4, 128, 'bf16', args.frames_per_gpu, std=False,
                                          kern_ms=timer_w.kernel_ms(n_w), clocks=clk_w.cpu().numpy().copy())
        roof_w.pop('_std'); roof_w.pop('_alg')
        f_train_w = mlp_flops(4, 128)[3]
        tfw = f_train_w * samples_step / dt / 1e12
        roof_w['step_algorithmic_tflops'] = round(tfw, 2)
        roof_w['step_mfma_frac'] = round(tfw / PEAK_TFLOPS['bf16'], 4)          # 322,560 flop/point at 4x128
        width128 = {'workload': 'config-2 geometry (%dx%d rays x %d samples, %d frames/step, loss full), 4x128 MLP' % (H, W, G, batch),
                    'dtype': 'bf16', 'ms_per_step': round(1e3 * dt, 3), 'value': round(samples_step / dt, 1), 'unit': 'ray-samples/s',
                    'steps': n_w, 'tape_frame_group': grp_w, 'roofline': roof_w, 'loss': float(torch.as_tensor(opt_w.loss).float().mean())}
        # the same steps with the step captured into a HIP graph (hparams['hip_graph']): same kernels, one launch
        train_step.use_graph = True
        try:
            run_steps(opt_w, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(opt_w, n_w)
            torch.cuda.synchronize()
            width128['ms_per_step_hip_graph'] = round(1e3 * (time.perf_counter() - t0) / n_w, 3)
        finally:
            train_step.use_graph = False
            train_step._graphs.clear()
        del opt_w, pred_w, eng_w, geom_w
        torch.cuda.empty_cache()
      except Exception as exc:
        width128 = {'error': repr(exc)}

    # ---- a network OUTSIDE the fused kernels' range (posenc_deg 5: csrc/general_mlp.hip, its bf16 kernels), same geometry ----
    general_path = None
    if world == 1 and std and not args.no_width128:
      try:
        pred_g = network.NeRF_Predictor(*dom, posenc_deg=5, net_depth=4, net_width=128, mode='bf16', device=dev)
        opt_g = optimization.Optimizer(hparams, pred_g, rt_args)
        run_steps(opt_g, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(opt_g, 4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        fg = mlp_flops(4, 128, F=33)[3]
        general_path = {'workload': 'config-2 geometry, 4x128 MLP with posenc_deg 5 (33 encoded inputs): the general layer-by-layer path, DESIGN.md 4.7',
                        'dtype': 'bf16 (v_mfma_f32_32x32x16_bf16 on fragment-ordered operands: gen_mlp16_kernel / gen_dw16_kernel, round 6)', 'ms_per_step': round(1e3 * dt, 3),
                        'value': round(samples_step / dt, 1), 'unit': 'ray-samples/s', 'steps': 4,
                        'step_algorithmic_tflops': round(fg * samples_step / dt / 1e12, 2),
                        'step_mfma_frac': round(fg * samples_step / dt / 1e12 / PEAK_TFLOPS['bf16'], 4), 'peak_tflops': PEAK_TFLOPS['bf16']}
        del opt_g, pred_g
        torch.cuda.empty_cache()
      except Exception as exc:
        general_path = {'error': repr(exc)}

    # ---- stand-alone radiative-transfer scan (kgeo.radiative_trasfer, HBM-bound): achieved GB/s ---------
    # measured at the size SURVEY 8d quotes (config 3: 256x256 rays x 128 samples, B*S = 8*3 planes, ~1 GB)
    from bhnerf_amd.kgeo import _RadiativeTransfer
    Nrt, Rrt, Grt = 24, 256 * 256, 128
    e_rt = torch.rand((Nrt, Rrt, Grt), device=dev)
    planes = [torch.rand((Rrt, Grt), device=dev) for _ in range(3)]
    rt_ms = timed(lambda: _RadiativeTransfer.apply(e_rt, *planes), reps=10)
    rt_bytes = 4 * Nrt * Rrt * Grt + 12 * Rrt * Grt + 4 * Nrt * Rrt                 # SURVEY 8d algorithmic bytes
    rt_scan = {'kernel': 'rt_kernel', 'bound': 'hbm', 'achieved': round(rt_bytes / (rt_ms * 1e-3) / 1e9, 1), 'peak': 8000.0,
               'unit': 'GB/s', 'frac': round(rt_bytes / (rt_ms * 1e-3) / 8e12, 4), 'ms': round(rt_ms, 4),
               'shape': '%d planes x %d rays x %d samples (%.2f GB)' % (Nrt, Rrt, Grt, rt_bytes / 1e9)}
    del e_rt, planes

    # ---- EHT visibility loss at BASELINE config 4's size (EHT2017: 28 baselines, 256x256 image, 8 frames): HBM-bound ----
    Ne, nvis, Re = 8, 28, 256 * 256
    img_e = torch.rand((Ne, Re), device=dev)
    A_e = torch.view_as_complex(torch.randn((Ne, nvis, Re, 2), device=dev))
    tgt_e = torch.view_as_complex(torch.randn((Ne, nvis, 2), device=dev))
    sig_e = torch.ones((Ne, nvis), device=dev)
    eht_ms = timed(lambda: engine.chi2_eht(img_e, A_e, tgt_e, sig_e, 1.0, 'vis'), reps=10)
    eht_bytes = 2 * 8 * Ne * nvis * Re                                          # A read by the GEMV and by its adjoint
    eht_loss = {'kernel': 'eht_vis_kernel + eht_loss_kernel + eht_bwd_kernel', 'bound': 'hbm', 'achieved': round(eht_bytes / (eht_ms * 1e-3) / 1e9, 1),
                'peak': 8000.0, 'unit': 'GB/s', 'frac': round(eht_bytes / (eht_ms * 1e-3) / 8e12, 4), 'ms': round(eht_ms, 4),
                'shape': "loss 'vis' + gradient, %d frames x %d visibilities x %d pixels (%.1f MB)" % (Ne, nvis, Re, eht_bytes / 1e6)}
    del img_e, A_e, tgt_e, sig_e

    # the same workload with the tutorials' recovery domain (rmin 2 M, rmax = fov/2, |z| <= 4 M): the samples outside it
    # have emission 0 and are compacted away (engine.RayGeometry.compact).  Reported next to the all-active headline,
    # never as `value`; single GPU only.
    tutorial_domain = None
    if world == 1 and not args.masked and not args.no_tutorial_domain:
        pred_m = network.NeRF_Predictor(rmax, 2.0, rmax, 4.0, net_depth=args.depth, net_width=args.width, mode=args.mode, device=dev)
        opt_m = optimization.Optimizer(hparams, pred_m, rt_args)
        run_steps(opt_m, max(args.warmup, 2))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(opt_m, args.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        gm = pred_m.geometry(rt_args['coords'], rt_args['Omega'], rt_args['t_geos'], None, rt_args['g'], rt_args['dtau'], rt_args['Sigma'])
        tutorial_domain = {'ms_per_step': round(1e3 * dt, 3), 'value': round(samples_step / dt, 1), 'unit': 'ray-samples/s',
                           'active_fraction': round(gm.active_fraction, 4), 'visited_fraction': round(gm.visited_fraction, 4)}
        del opt_m, pred_m

    out = {
        'metric': 'train ray-samples/sec, 128x128x64-sample image-plane recovery', 'value': value,
        'unit': 'ray-samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.mode, 'data': 'synthetic',
        'config': {'workload': 'Tutorial3 image-plane recovery: %dx%d rays x %d samples, %d frames, %dx%d MLP, loss full'
                               % (H, W, G, nt, args.depth, args.width),
                   'frames_per_step': batch, 'frames_per_gpu': args.frames_per_gpu, 'parallelism': 'dp%d (time-frames)' % world + (', process group %s' % backend if backend else '') + (', stale-gradient all-reduce overlap' if (args.overlap_allreduce and world > 1) else ''),
                   'active_fraction': round(geom.active_fraction, 4), 'visited_fraction': round(geom.visited_fraction, 4),
                   'tape_frame_group': group,
                   'loss': loss_now},
        'roofline': roofline,
        'rt_scan': rt_scan,
        'eht_loss': eht_loss,
        'fwd_images_per_s': fwd_path['value'] if fwd_path else None,
        'fwd_path': fwd_path,
    }
    if width128:
        out['width128'] = width128
    if general_path:
        out['general_path'] = general_path
    if world == 1 and std and args.other_configs:
        try:
            out['strong_scaling_share'] = strong_scaling_share(dev, args.mode)
        except Exception as exc:
            out['strong_scaling_share'] = {'error': repr(exc)}
    if args.other_configs and world == 1 and args.mode == 'bf16' and std:
        del opt
        torch.cuda.empty_cache()
        try:
            out['other_configs'] = other_configs(dev, args.mode)
        except Exception as exc:                     # (an OOM or a missing fixture here must not cost the headline line)
            out['other_configs'] = {'error': repr(exc)}
    if parity:
        out['parity_mode'] = parity
    if tape8:
        out['tape8_mode'] = tape8
    if tutorial_domain:
        out['tutorial_domain'] = tutorial_domain
    if cpu:
        out['cpu_baseline'] = cpu
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
