"""Host-side cost of one training step (cProfile) at a small per-GPU workload (BASELINE config 5 with one frame per GPU:
64 x 64 rays x 100 samples, 4x128, 'lc'), where the Python driver, not the GPU, sets the step time.
    python tools/prof_host_step.py [frames_per_step]"""
import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bhnerf_amd import network, optimization, synthetic, units
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda:0')
geo = synthetic.synthetic_geodesics(64, 64, 100, fov_M=40.0, inc_deg=12.0, S=3, seed=3)
t_frames = np.linspace(0.0, 1.7, 128)
pred = network.NeRF_Predictor(20.0, 6.0, 20.0, 4.0, net_depth=4, net_width=128, mode='bf16', device=dev)
rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                  t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=geo['J'])
target = np.random.default_rng(5).uniform(0.5, 1.5, (128, 3)).astype(np.float32)
step = optimization.TrainStep.image(t_frames * units.hr, target, sigma=0.1, dtype='lc')
opt = optimization.Optimizer({'num_iters': 100000, 'lr_init': 1e-4, 'lr_final': 1e-6}, pred, rt)
frames = step.args[0]
def run(n):
    for _ in range(n):
        opt.loss, opt.state, _ = step(opt.state, rt, indices=frames.sample(B))
run(50); torch.cuda.synchronize()
t0 = time.perf_counter(); run(1000); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 1000
print('frames/step %d: %.1f us per step end to end' % (B, dt * 1e6))
pr = cProfile.Profile(); pr.enable(); run(1000); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
