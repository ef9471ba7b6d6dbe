"""dW kernel time as a function of the number of workgroups (BHN_DEBUG_DW_GRID): is the tape stream bound per CU or by HBM?"""
import os; os.environ.setdefault('BHNERF_HIP_LIB', '/root/repo/bhnerf_amd/csrc/libbhnerf_hip_dbg.so')
import os, subprocess, sys
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bhnerf_amd import _hip, engine, network, synthetic, constants
    dev = torch.device('cuda:0')
    H = W = 128; G = 64; B = 8
    geo = synthetic.synthetic_geodesics(H, W, G)
    WID, DEP = int(os.environ.get('WIDTH', 256)), int(os.environ.get('DEPTH', 4))
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=DEP, net_width=WID, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    flat = eng.flatten(network.MLP(DEP, WID).init(1, 21)); eng.pack(flat)
    tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
    def timed(fn, reps=4):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))
    lib = _hip.lib()
    lib.bhn_debug_set_bwd_stages(7); eng.render_bwd(geom, tM0, dimg)
    for name, mask in [('chain', 1), ('dw', 2)]:
        lib.bhn_debug_set_bwd_stages(mask)
        print('grid', os.environ.get('BHN_DEBUG_DW_GRID', 'ncu'), name, '%.3f ms' % timed(lambda: eng.render_bwd(geom, tM0, dimg)), flush=True)
else:
    if os.environ.get('SWEEP') == 'jobs':
        for jl in os.environ.get('JL', '1.5 4 6 8 10 12').split():
            for j1 in os.environ.get('J1', '12 14').split():
                print('JOBL_W', jl, 'JOB1_W', j1, flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), 'run'], env=dict(os.environ, BHN_DEBUG_JOBL_W=jl, BHN_DEBUG_JOBLB_W=jl, BHN_DEBUG_JOB1_W=j1))      # (JL sets both the h-tile and the relu-bit form of the layer depth-1 job)
    else:
        for g in ['0', '224', '192', '160', '128', '96', '64']:
            subprocess.run([sys.executable, os.path.abspath(__file__), 'run'], env=dict(os.environ, BHN_DEBUG_DW_GRID=g))
