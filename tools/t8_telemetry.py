"""Board power / SMI clock while each kernel of the 4x256 step runs back to back for a few seconds, bf16 against the 8-bit tape
mode (run beside tools/smi_sample.py; prints the window of every row, tools/r4_job15.sh joins them):
    python3 tools/t8_telemetry.py [seconds per row]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhnerf_amd import engine, network, synthetic, constants

dev = torch.device('cuda:0')
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
geo = synthetic.synthetic_geodesics(128, 128, 64, seed=0)
for mode in ('bf16', 'bf16_t8'):
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=256, mode=mode, device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    eng.pack(eng.flatten(network.MLP(4, 256).init(1, 21)))
    tM0 = engine.frame_offsets(np.linspace(0, 1, 8), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = torch.rand((8, 1, geom.R), device=dev) * 1e-3
    eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg); torch.cuda.synchronize()
    for name, fn in (('fwd_train', lambda: eng.render_train(geom, tM0)), ('backward', lambda: eng.render_bwd_tape(geom, tM0, dimg)),
                     ('inference', lambda: eng.render(geom, tM0))):
        time.sleep(1.0)
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize(); n += 20
        t1 = time.time()
        print('ROW %s %s %.2f %.2f %d %.3f' % (mode, name, t0, t1, n, 1e3 * (t1 - t0) / n), flush=True)
