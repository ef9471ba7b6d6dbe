import torch, numpy as np
dev = torch.device('cuda:0')
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
n = 16 * 2**30
x = torch.empty(n // 4, dtype=torch.float32, device=dev)
y = torch.empty(n // 4, dtype=torch.float32, device=dev)
t = timed(lambda: x.fill_(1.0)); print('fill 16 GiB: %.2f ms -> %.2f TB/s write' % (t, n / t / 1e9))
t = timed(lambda: y.copy_(x)); print('copy 16 GiB: %.2f ms -> %.2f TB/s (r+w)' % (t, 2 * n / t / 1e9))
t = timed(lambda: x.sum()); print('sum 16 GiB: %.2f ms -> %.2f TB/s read' % (t, n / t / 1e9))
