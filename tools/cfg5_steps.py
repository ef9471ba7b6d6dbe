"""BASELINE config 5's step (64x64 rays x 100 samples, Stokes I/Q/U, loss lc, 4x128, tutorial-style domain) in a loop, for
rocprofv3 --kernel-trace --stats:  python3 tools/cfg5_steps.py [frames_per_step] [steps] [graph]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhnerf_amd import network, optimization, synthetic, units
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'
dev = torch.device('cuda:0')
c = dict(H=64, W=64, G=100, width=128, fov=40.0, inc=12.0, spin=0.0, rmin=6.0, rmax=20.0, z_width=4.0)
geo = synthetic.synthetic_geodesics(c['H'], c['W'], c['G'], fov_M=c['fov'], inc_deg=c['inc'], spin=c['spin'], S=3, seed=3)
nt = 128
t_frames = np.linspace(0.0, 1.7, nt)
rt = network.raytracing_args(dict(x=geo['coords'][0], y=geo['coords'][1], z=geo['coords'][2], dtau=geo['dtau'], Sigma=geo['Sigma'],
                                  t=geo['t_geos'], g=geo['g']), geo['Omega'], geo['t_injection'], 0.0 * units.hr, J=geo['J'])
target = np.random.default_rng(5).uniform(0.5, 1.5, (nt, 3)).astype(np.float32)
pred = network.NeRF_Predictor(c['rmax'], c['rmin'], c['rmax'], c['z_width'], net_depth=4, net_width=c['width'], mode='bf16', device=dev)
step = optimization.TrainStep.image(t_frames * units.hr, target, sigma=0.1, dtype='lc')
step.use_graph = graph
opt = optimization.Optimizer({'num_iters': 100000, 'lr_init': 1e-4, 'lr_final': 1e-6}, pred, rt)
frames = step.args[0]
for _ in range(5):
    opt.loss, opt.state, _ = step(opt.state, rt, frames.sample(B))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    opt.loss, opt.state, _ = step(opt.state, rt, frames.sample(B))
torch.cuda.synchronize()
print('config 5, %d frames/step, %s: %.4f ms/step' % (B, 'graph' if graph else 'eager', 1e3 * (time.perf_counter() - t0) / N))
