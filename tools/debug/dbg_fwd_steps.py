"""Per-ring-step time stamps of one tile of the 4x256 bf16 render kernel (ablation build, flag 128):
for waves 0 and 4 of workgroup 0: cycles spent computing, waiting for the weight DMA, waiting at the barrier."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
import os; os.environ.setdefault('BHNERF_HIP_LIB', '/root/repo/bhnerf_amd/csrc/libbhnerf_hip_dbg.so')   # debug build: make -C bhnerf_amd/csrc debug
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')
lib = _hip.lib()
width, H, G, B = 256, 128, 64, 8
geo = synthetic.synthetic_geodesics(H, H, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=width, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, width).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
for extra, name in ((64, 'full'), (64 | 1, 'no MFMAs')):
    lib.bhn_debug_set_fwd_variant(3 | ((128 | extra) << 4))
    for _ in range(3):
        eng.render(geom, tM0)
    torch.cuda.synchronize()
    buf = np.zeros(512, dtype=np.int64)
    _hip.check(lib.bhn_debug_read(buf.ctypes.data_as(C.c_void_p), 4096))
    t = buf.reshape(8, 32, 2)[:, :26]          # wave, step, (compute done, barrier passed)
    t0 = t[:, 0, 1].min()
    print(name, ': tile', t[0, -1, 1] - t[0, 0, 1], 'ticks for steps 1..25')
    print('  step: compute ticks per wave 0..7 | barrier wait per wave')
    for k in range(1, 26):
        comp = t[:, k, 0] - t[:, k - 1, 1]
        wait = t[:, k, 1] - t[:, k, 0]
        print('  %2d: %s | %s' % (k, ' '.join('%5d' % c for c in comp), ' '.join('%4d' % w for w in wait)))
    print('  step 0 arrival spread (compute-done time of each wave minus earliest):', ' '.join('%d' % d for d in (t[:, 0, 0] - t[:, 0, 0].min())))
lib.bhn_debug_set_fwd_variant(1)
