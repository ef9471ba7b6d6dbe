"""Image-plane recovery end to end on one MI355X (the flow of the reference's Tutorial 3), with nothing but this package:

  Kerr geodesics (own tracer)  ->  Doppler factor  ->  a rotating Gaussian hotspot rendered through the voxel
  renderer as the 'observed' movie  ->  NeRF_Predictor trained on the image-plane chi^2  ->  3-D emission sampled back.

    python examples/image_plane_recovery.py [--size 32] [--ngeo 48] [--iters 300]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhnerf_amd as bhnerf  # noqa: E402
from bhnerf_amd import kgeo, network, optimization, units  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=32)
    ap.add_argument('--ngeo', type=int, default=48)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--width', type=int, default=128)
    ap.add_argument('--mode', default='bf16', choices=['bf16', 'f32'], help='arithmetic of the fused kernels')
    ap.add_argument('--batch', type=int, default=4, help='frames per step')
    args = ap.parse_args()
    fov, rmax = 16.0, 8.0
    geos = kgeo.image_plane_geos(0.3, np.deg2rad(30.0), (-fov / 2, fov / 2), (-fov / 2, fov / 2), ngeo=args.ngeo,
                                 num_alpha=args.size, num_beta=args.size)
    Omega = np.nan_to_num(1.0 / (np.sqrt(geos.x ** 2 + geos.y ** 2) ** 1.5 + geos.spin))      # Keplerian about the spin axis
    geos['g'] = kgeo.doppler_factor(geos, kgeo.azimuthal_velocity_vector(geos, Omega))
    # ground truth: a Gaussian hotspot on a 32^3 grid, orbiting with Omega
    n = 32
    ax = np.linspace(-rmax, rmax, n)
    gx, gy, gz = np.meshgrid(ax, ax, ax, indexing='ij')
    hotspot = np.exp(-((gx - 5.0) ** 2 + gy ** 2 + gz ** 2) / (2 * 0.8 ** 2))
    t_frames = np.linspace(0.0, 1.5, args.frames) * units.hr
    t_injection = -float(geos.r_o)
    movie = bhnerf.emission.image_plane_dynamics((hotspot, 2 * rmax), geos, Omega, t_frames, t_injection, J=1.0, doppler=True)
    print('observed movie', movie.shape, 'flux range %.3g .. %.3g' % (movie.sum((-1, -2)).min(), movie.sum((-1, -2)).max()))

    rt = network.raytracing_args(geos, Omega, t_injection, t_frames[0], J=1.0)
    predictor = network.NeRF_Predictor(rmax, 2.0, rmax, 4.0, net_depth=4, net_width=args.width, mode=args.mode)
    train_step = optimization.TrainStep.image(t_frames, movie, sigma=float(movie.max()) * 0.05, dtype='full')
    opt = optimization.Optimizer({'num_iters': args.iters, 'lr_init': 1e-3, 'lr_final': 1e-4}, predictor, rt)
    import time
    import torch
    first = optimization.total_movie_loss(args.batch, opt.state, train_step, rt)
    t0 = time.perf_counter()
    opt.run(args.batch, train_step, rt, log_fns=[optimization.LogFn(lambda o: print('iter %4d  chi2/frame %.4g' % (o.step, float(np.mean(o.loss)))), max(50, args.iters // 10))])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last = optimization.total_movie_loss(args.batch, opt.state, train_step, rt)
    print('movie chi2 %.4g -> %.4g   (%d iterations in %.1f s = %.1f ms/iteration, %.3g ray-samples/s)' % (
        first, last, args.iters, dt, 1e3 * dt / args.iters, args.iters * args.batch * args.size ** 2 * args.ngeo / dt))
    vol = network.sample_3d_grid(predictor.apply, opt.state.params, fov=2 * rmax, resolution=n)
    i = np.unravel_index(np.argmax(vol), vol.shape)
    print('recovered emission peaks at (x, y, z) = (%.1f, %.1f, %.1f) M; truth (5.0, 0.0, 0.0) at t = 0' % (ax[i[0]], ax[i[1]], ax[i[2]]))


if __name__ == '__main__':
    main()
