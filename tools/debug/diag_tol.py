import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import golden_tree
import test_gpu_backward as tb
import test_gpu_forward as tf
from bhnerf_amd import network, units
dev = torch.device('cuda:0')
G = lambda n: dict(np.load('/root/repo/tests/golden/%s.npz' % n))
for tag in tb.PRED:
    g = G('g5_predict_' + tag)
    for mode in ('f32', 'bf16'):
        # forward
        pred, tree = tf._predictor(g, mode, dev)
        e = pred.apply({'params': tree}, g['t_frames'], units.hr, g['coords'].astype(np.float32), g['Omega'].astype(np.float32),
                       float(g['t_start_obs']), g['t_geos'].astype(np.float32), float(g['t_injection'])).cpu().numpy()
        mism = ((e == 0) != (g['emission'] == 0))
        same = ~mism
        f = lambda k: g[k].astype(np.float32)
        J = f('J') if g['J'].ndim else 1.0
        with torch.no_grad():
            images = network.image_plane_prediction(tree, pred.apply, g['t_frames'], f('coords'), f('Omega'), J, f('g'), f('dtau'), f('Sigma'), float(g['t_start_obs']), f('t_geos'), float(g['t_injection']), units.hr).cpu().numpy()
        line = '%s %-4s W=%d mism %d/%d  e_err %.2e  img_err %.2e' % (tag, mode, int(g['hparams'][6]), mism.sum(), mism.size, tf.relerr(e[same], g['emission'][same]), tf.relerr(images, g['images']))
        for dt in ('full', 'lc'):
            tr, t = tb.oracle_trainer(g)
            tg = tb.targets(g, dt)
            scale = float(g['hparams'][7])
            loss_ref, _, grads_ref = tr.loss_and_grad(t(g['t_frames']), t(tg['target']), t(tg['sigma']), t(tg['offset']), scale, dt)
            n = len(tr.k)
            gref = np.concatenate([np.concatenate([grads_ref[i].numpy().ravel(), grads_ref[n + i].numpy().ravel()]) for i in range(n)])
            pred2, rt = tb.device_setup(g, mode, dev)
            params = pred2.engine().flatten(golden_tree(g)).requires_grad_(True)
            tree2 = network.ParamTree(); tree2.flat = params
            loss, [im] = network.loss_fn_image(tree2, pred2.apply, tg['target'], tg['sigma'], tg['offset'], g['t_frames'], rt['coords'], rt['Omega'], rt['J'], rt['g'], rt['dtau'], rt['Sigma'], rt['t_start_obs'], rt['t_geos'], rt['t_injection'], scale, units.hr, dt)
            loss.backward()
            gdev = params.grad.cpu().numpy().astype(np.float64)
            line += '  | %s: L2 %.2e max %.2e loss %.1e' % (dt, tb.l2err(gdev, gref), np.abs(gdev - gref).max() / np.abs(gref).max(), abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()))
        print(line, flush=True)
