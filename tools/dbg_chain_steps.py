"""Per-ring-step time stamps (all 8 waves, one tile of workgroup 0) of the training-forward and delta-chain kernels
at config 2: ticks spent computing and ticks waiting (DMA wait + barrier) per step.  bhn_debug_set_bwd_stages bit 12.
Needs the library built with the stamps compiled in:  make -C bhnerf_amd/csrc CXXFLAGS="... -DBHN_CHAIN_STAMPS=1"."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import os; os.environ.setdefault('BHNERF_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bhnerf_amd/csrc/libbhnerf_hip_dbg.so'))   # debug build: make -C bhnerf_amd/csrc debug
from bhnerf_amd import _hip, engine, network, synthetic, constants
dev = torch.device('cuda:0')
lib = _hip.lib()
H = W = 128; G = 64; B = 8
geo = synthetic.synthetic_geodesics(H, W, G)
pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=256, mode='bf16', device=dev)
eng = pred.engine()
geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
eng.pack(eng.flatten(network.MLP(4, 256).init(1, 21)))
tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg)
def show(name, nsteps):
    torch.cuda.synchronize()
    buf = np.zeros(512, dtype=np.int64)
    _hip.check(lib.bhn_debug_read(buf.ctypes.data_as(C.c_void_p), 4096))
    t = buf.reshape(8, 32, 2)[:, :nsteps]
    print(name, ': steps 1..%d take' % (nsteps - 1), t[0, -1, 1] - t[0, 0, 1], 'ticks')
    for k in range(1, nsteps):
        comp = t[:, k, 0] - t[:, k - 1, 1]
        wait = t[:, k, 1] - t[:, k, 0]
        print('  %2d: %s | %s' % (k, ' '.join('%5d' % c for c in comp), ' '.join('%4d' % w for w in wait)))
    print('  step 0 compute-done spread:', ' '.join('%d' % d for d in (t[:, 0, 0] - t[:, 0, 0].min())))
for extra, tag in ((0, ''), (1 << 9, ' (emit w/o global stores)'), (1 << 10, ' (no emit)')):
    lib.bhn_debug_set_bwd_stages(7 | (1 << 12) | extra)
    eng.render_train(geom, tM0)
    show('training forward' + tag, 26)
    lib.bhn_debug_set_bwd_stages(1 | (1 << 12) | extra)
    eng.render_bwd_tape(geom, tM0, dimg)
    show('delta chain' + tag, 24)
lib.bhn_debug_set_bwd_stages(7)
