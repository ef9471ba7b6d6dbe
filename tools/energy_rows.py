"""Energy budget of the training step (round 4): every kernel of the step, and isolated parts of the dW
kernel (debug build), run back to back for S seconds each while tools/smi_sample.py samples the board beside it; the
windows are joined with the telemetry afterwards (mode `join`).

    python3 tools/energy_rows.py run <seconds> [width]  >> rows.txt        (release or debug library via BHNERF_HIP_LIB)
    python3 tools/energy_rows.py join telemetry.txt rows.txt [rows2.txt ...]

Row format (same as tools/energy_bench.hip / step_bench.hip): name, `unix t0 .. t1`, launches, ms each."""
import os, re, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(sec, width):
    import torch
    from bhnerf_amd import _hip, engine, network, synthetic, constants
    dev = torch.device('cuda:0')
    H = W = 128; G = 64; B = 8
    geo = synthetic.synthetic_geodesics(H, W, G)
    pred = network.NeRF_Predictor(8.0, 0.0, np.inf, np.inf, net_depth=4, net_width=width, mode='bf16', device=dev)
    eng = pred.engine()
    geom = pred.geometry(geo['coords'], geo['Omega'], geo['t_geos'], None, geo['g'], geo['dtau'], geo['Sigma'])
    flat = eng.flatten(network.MLP(4, width).init(1, 21)); eng.pack(flat)
    tM0 = engine.frame_offsets(np.linspace(0, 1, B), 0.0, geo['t_injection'], constants.GM_c3('hr'), dev)
    dimg = torch.rand((B, 1, geom.R), device=dev) * 1e-3
    lib = _hip.lib()
    dbg = hasattr(lib, 'bhn_debug_set_bwd_stages') and 'dbg' in os.environ.get('BHNERF_HIP_LIB', '')
    pts = B * geom.P

    def loop(name, fn, per_launch=1):
        fn(); torch.cuda.synchronize()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = 0.0
        while time.time() - t0 < sec:
            e0.record()
            for _ in range(8):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms += e0.elapsed_time(e1); n += 8
        t1 = time.time()
        print('%-28s unix %.1f .. %.1f  %4d launches %8.3f ms each  width %d  points %d' % (name, t0, t1, n, ms / n, width, pts), flush=True)

    eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg)
    if not dbg:
        loop('inference_fwd', lambda: eng.render(geom, tM0))
        loop('train_fwd', lambda: eng.render_train(geom, tM0))
        loop('bwd(chain+dw+reduce)', lambda: eng.render_bwd_tape(geom, tM0, dimg))

        def step():
            eng.render_train(geom, tM0); eng.render_bwd_tape(geom, tM0, dimg)
        loop('fwd+bwd', step)
    else:
        rt = lambda: eng.render_bwd_tape(geom, tM0, dimg)
        for name, mask in [('chain_only', 1), ('dw_full', 2), ('dw_stream_only(no mfma)', 2 | 8), ('dw_compute_only(no loads)', 2 | 16)]:
            lib.bhn_debug_set_bwd_stages(mask)
            loop(name, rt)
        lib.bhn_debug_set_bwd_stages(7)


def join(tele, rows):
    T = []
    for l in open(tele):
        if l.startswith('#'):
            continue
        p = l.split()
        if len(p) >= 3:
            T.append((float(p[0]), float(p[1]), float(p[2])))
    T = np.array(T)
    idle = np.percentile(T[:, 1], 2)
    print('# idle board power (2nd percentile of all samples): %.0f W' % idle)
    print('%-44s %8s %8s %9s %9s %9s  %s' % ('row', 'W', 'SMI MHz', 'ms', 'J/launch', 'J-idle', 'rest of the row'))
    for f in rows:
        for l in open(f):
            m = re.match(r'(.*?)\s+unix ([\d.]+) \.\. ([\d.]+)\s+(\d+) launches\s+([\d.]+) ms each(.*)', l)
            if not m:
                continue
            name, t0, t1, n, ms, rest = m.group(1), float(m.group(2)), float(m.group(3)), int(m.group(4)), float(m.group(5)), m.group(6)
            sel = T[(T[:, 0] > t0 + 0.6) & (T[:, 0] < t1 - 0.3)]
            if len(sel) == 0:
                print('%-44s (no telemetry samples)' % name)
                continue
            w, mhz = sel[:, 1].mean(), sel[:, 2].mean()
            busy = n * ms * 1e-3 / (t1 - t0)                 # fraction of the window the GPU was running the row
            wk = idle + (w - idle) / max(busy, 1e-3)           # power while the kernel runs (host gaps draw idle power)
            print('%-44s %8.0f %8.0f %9.3f %9.3f %9.3f  busy %.2f%s' % (name, wk, mhz, ms, wk * ms * 1e-3, (wk - idle) * ms * 1e-3, busy, rest.rstrip()))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 256)
    else:
        join(sys.argv[2], sys.argv[3:])
