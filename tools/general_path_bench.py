"""Training step and render of network shapes OUTSIDE the fused kernels (csrc/general_mlp.hip: posenc_deg > 4, net_width > 256) on
BASELINE config 2's problem (128x128 rays x 64 samples, all-active, loss full), beside the fused 4x256 / 4x128 of the same problem:
    python3 tools/general_path_bench.py [frames_per_step] [steps]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhnerf_amd import network, optimization, synthetic, units
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
H = W = 128; G = 64
geo = synthetic.synthetic_geodesics(H, W, G, fov_M=16.0, inc_deg=60.0, seed=3)
nt = 64
t_frames = np.linspace(0.0, 1.0, nt)
rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                  t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=1.0)
target = np.random.default_rng(5).uniform(0.0, 1e-3, (nt, H, W)).astype(np.float32)
ONLY = os.environ.get('GEN_ONLY')          # e.g. GEN_ONLY=2: only the third shape of the list (for rocprofv3 --stats)
for ci, (depth, width, deg, mode) in enumerate(((4, 256, 3, 'bf16'), (4, 128, 5, 'bf16'), (4, 256, 6, 'bf16'), (4, 512, 3, 'bf16'), (8, 512, 10, 'bf16'),
                                                (4, 256, 3, 'f32'), (4, 128, 5, 'f32'), (4, 256, 6, 'f32'), (4, 512, 3, 'f32'), (8, 512, 10, 'f32'))):
    if ONLY is not None and int(ONLY) != ci:
        continue
    pred = network.NeRF_Predictor(16.0, 0.0, np.inf, np.inf, posenc_deg=deg, net_depth=depth, net_width=width, mode=mode, device=dev)
    step = optimization.TrainStep.image(t_frames * units.hr, target, sigma=1.0, dtype='full')
    opt = optimization.Optimizer({'num_iters': 100000, 'lr_init': 1e-4, 'lr_final': 1e-6}, pred, rt)
    frames = step.args[0]
    for _ in range(2):
        opt.loss, opt.state, _ = step(opt.state, rt, frames.sample(B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        opt.loss, opt.state, _ = step(opt.state, rt, frames.sample(B))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / N
    pts = B * H * W * G
    fl = 2.0 * (3 + 6 * deg) * width + 2.0 * (depth - 1) * width * width + 2.0 * width + (2.0 * (3 + 6 * deg) * width if depth >= 2 else 0)
    print('%dx%d posenc %2d %-4s %s: %9.2f ms/step  %7.1f M ray-samples/s  ~%6.1f TFLOP/s (3 x forward flops)' %
          (depth, width, deg, mode, 'general' if (deg > 4 or width > 256) else 'fused  ', ms, pts / ms / 1e3, 3 * fl * pts / ms / 1e9))
