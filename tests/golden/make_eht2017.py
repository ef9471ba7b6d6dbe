#!/usr/bin/env python3
"""Fixture for BASELINE config 4 (Tutorial4: EHT2017 array, visibility-domain loss): earth-rotation (u, v) tracks of the
EHT2017 stations on Sgr A* for the 64 frames of a recovery.

    python3 tests/golden/make_eht2017.py          (build container only: reads the reference's data file)

The station coordinates and SEFDs are DATA of the reference (`/root/reference/eht_arrays/EHT2017.txt`, the file
Tutorial4 hands to ehtim).  ehtim itself (the reference's `observation.py` / `TrainStep.eht`, optimization.py:219-268)
is an absent third-party package, so the (u, v) synthesis is restated here from the textbook relation (Thompson, Moran &
Swenson eq. 4.1) -- parity with ehtim's own scan bookkeeping is NOT pinned; what the test checks downstream is
`loss_fn_eht` given these A matrices.  Only data is written: tests/golden/g11_eht2017.npz
    sites (8,) str, xyz (8,3) m, sefd (8,) Jy, pairs (28,2), t_hr (64,), uv (64,28,2) wavelengths at 230 GHz,
    up (64,28) bool (both stations see the source above 10 deg), sigma (64,28) Jy thermal noise.
"""
import os

import numpy as np

SRC = '/root/reference/eht_arrays/EHT2017.txt'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g11_eht2017.npz')

names, xyz, sefd = [], [], []
for line in open(SRC):
    if line.startswith('#') or not line.strip():
        continue
    f = line.split()
    names.append(f[0]); xyz.append([float(v) for v in f[1:4]]); sefd.append(0.5 * (float(f[4]) + float(f[5])))
xyz, sefd = np.array(xyz), np.array(sefd)
n = len(names)
pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)])

ra = (17 + 45 / 60 + 40.0409 / 3600) * 15.0 * np.pi / 180          # Sgr A* J2000
dec = -(29 + 0 / 60 + 28.118 / 3600) * np.pi / 180
lam = 299792458.0 / 230e9
nt = 64
t_hr = np.linspace(0.0, 8.0, nt)                                       # an 8 h track, one frame every 7.6 min
gst0 = ra + np.deg2rad(110.0) - 4.0 / 24 * 2 * np.pi                   # mid-track the source transits longitude 110 W (between Chile and Hawaii)
gst = gst0 + t_hr / 23.9344696 * 2 * np.pi
H = gst - ra                                                           # Greenwich hour angle of the source
B = xyz[pairs[:, 1]] - xyz[pairs[:, 0]]                                # baseline vectors, earth-fixed (m)
sH, cH = np.sin(H)[:, None], np.cos(H)[:, None]
u = (sH * B[None, :, 0] + cH * B[None, :, 1]) / lam
v = (-np.sin(dec) * cH * B[None, :, 0] + np.sin(dec) * sH * B[None, :, 1] + np.cos(dec) * B[None, :, 2]) / lam
uv = np.stack([u, v], axis=-1)
# elevation of the source at each station: source direction in the earth-fixed frame (hour angle H west of Greenwich)
s = np.stack([np.cos(dec) * np.cos(-H), np.cos(dec) * np.sin(-H), np.full_like(H, np.sin(dec))], axis=-1)      # (nt,3)
rhat = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
el = np.arcsin(np.clip(s @ rhat.T, -1, 1))                             # (nt, n)
vis_ok = el > np.deg2rad(10.0)
up = vis_ok[:, pairs[:, 0]] & vis_ok[:, pairs[:, 1]]
bw, tint = 4e9, 60.0                                                   # thermal noise of one scan (2-bit efficiency 0.88)
sigma = np.sqrt(sefd[pairs[:, 0]] * sefd[pairs[:, 1]]) / (0.88 * np.sqrt(2 * bw * tint))
sigma = np.broadcast_to(sigma, (nt, len(pairs))).copy()
np.savez_compressed(OUT, sites=np.array(names), xyz=xyz, sefd=sefd, pairs=pairs, t_hr=t_hr, uv=uv, up=up, sigma=sigma)
print('wrote', OUT, 'baselines', len(pairs), 'up fraction %.2f' % up.mean(), 'max |uv| %.2f Glambda' % (np.abs(uv).max() / 1e9))
