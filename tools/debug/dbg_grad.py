import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_backward as T
from bhnerf_amd import network, units
from oracle import oracle_np as onp
from conftest import golden_tree
dev = torch.device('cuda:0')
def run(width, depth, S, H=9, Wd=7, G=50, B=3):
    rng = np.random.default_rng(width + depth)
    alpha, beta = np.meshgrid(np.linspace(-8, 8, H), np.linspace(-8, 8, Wd), indexing='ij')
    s = np.linspace(-9.6, 9.6, G); inc = np.deg2rad(60.0)
    coords = np.stack([alpha[..., None] * np.ones(G), beta[..., None] * np.cos(inc) + s * np.sin(inc), -beta[..., None] * np.sin(inc) + s * np.cos(inc)])
    r = np.sqrt((coords ** 2).sum(0)) + 0.3
    geo = dict(coords=coords, Omega=1.0 / (r ** 1.5 + 0.1), t_geos=-(1000.0 - (s + 9.6)) * np.ones_like(r), g=rng.uniform(0.6, 1.4, r.shape), Sigma=r ** 2, dtau=(s[1] - s[0]) / r ** 2)
    J = None
    if S:
        I = rng.uniform(0.5, 1.5, r.shape); chi = rng.uniform(0, np.pi, r.shape)
        J = np.stack([I, 0.85 * I * np.cos(2 * chi), 0.85 * I * np.sin(2 * chi)])[:S]
    t_frames = np.sort(rng.uniform(0, 1, B)); t_inj = -(1000.0 - 4.0)
    f32r = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
    geo = {k: f32r(v) for k, v in geo.items()}; J = f32r(J) if S else None
    tree = onp.he_uniform_params(rng, depth, width, 21, dtype=np.float32)
    for i in range(depth + 1):
        d = tree['MLP_0']['Dense_%d' % i]; d['kernel'] = d['kernel'].astype(np.float64); d['bias'] = f32r(rng.uniform(-0.1, 0.1, d['bias'].shape))
    g = dict(geo, J=(J if S else np.array(1.0)), t_frames=t_frames, t_start_obs=0.0, t_injection=t_inj, hparams=np.array([8.0, 2.5, 8.0, 4.0, 3, depth, width, 1.0]))
    for i in range(depth + 1):
        g['kernel%d' % i] = tree['MLP_0']['Dense_%d' % i]['kernel']; g['bias%d' % i] = tree['MLP_0']['Dense_%d' % i]['bias']
    tr, t = T.oracle_trainer(g)
    shape = (B, S, H, Wd) if S else (B, H, Wd)
    target = rng.uniform(0, 1e-3, shape); sigma = rng.uniform(0.5, 2.0, shape); offset = np.zeros(shape)
    loss_ref, img_ref, grads_ref = tr.loss_and_grad(t(t_frames), t(target), t(sigma), t(offset), 1.0, 'full')
    n = len(tr.k)
    pred, rt = T.device_setup(g, 'bf16', dev)
    eng = pred.engine()
    params = eng.flatten(golden_tree(g)).requires_grad_(True)
    ptree = network.ParamTree(); ptree.flat = params
    loss, [images] = network.loss_fn_image(ptree, pred.apply, target, sigma, offset, t_frames, rt['coords'], rt['Omega'], rt['J'], rt['g'], rt['dtau'], rt['Sigma'], 0.0, rt['t_geos'], t_inj, 1.0, units.hr, 'full')
    loss.backward()
    gd = params.grad.cpu().numpy()
    gmax = max(float(x.abs().max()) for x in grads_ref)
    for i in range(n):
        k = gd[eng.kernel_off[i]:eng.kernel_off[i] + eng.in_dim[i] * eng.out_dim[i]].reshape(eng.in_dim[i], eng.out_dim[i])
        b = gd[eng.bias_off[i]:eng.bias_off[i] + eng.out_dim[i]]
        ek = np.abs(k - grads_ref[i].numpy()); eb = np.abs(b - grads_ref[n + i].numpy())
        w = np.unravel_index(ek.argmax(), ek.shape)
        print('W%d D%d S%d layer %d: kernel err %.2e (at %s, ref %.3e dev %.3e) bias err %.2e | rows-err max by in-block: %s' % (
            width, depth, S, i, ek.max() / gmax, w, grads_ref[i].numpy()[w], k[w], eb.max() / gmax,
            ['%.1e' % (ek[j:j + 32].max() / gmax) for j in range(0, ek.shape[0], 32)]))
import sys as _s
MODE = 'bf16'
run(128, 4, 3); run(256, 4, 0)
