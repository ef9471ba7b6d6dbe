"""One-frame-per-GPU share (bench.strong_scaling_share) with and without a process group of ONE rank over RCCL, and with the
all-reduce captured inside the HIP graph (default) or left eager (BHNERF_GRAPH_COLLECTIVE=0):
    python3 tools/dbg_graph_dist.py [nodist|dist]"""
import json, os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
if (sys.argv[1:] or ['dist'])[0] == 'dist':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
out = bench.strong_scaling_share(dev, 'bf16')
print(json.dumps({'argv': sys.argv[1:], 'collective_in_graph': os.environ.get('BHNERF_GRAPH_COLLECTIVE', '1'), 'share': out}))
