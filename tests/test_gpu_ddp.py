"""Data-parallel step with the real kernels (network.py:566-622, optimization.py:163-179, 289-291): two ranks,
both on cuda:0 over gloo, a LIST of three ray sets.  Also `bench.py --gpus 2` as the driver invokes it."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.fixture(scope='module', autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')


@pytest.mark.timeout(600)
@pytest.mark.parametrize('overlap', ['0', '1'])        # '1': opt-in stale-gradient overlap of the all-reduce
def test_two_ranks_equal_single_process_mean_of_sums(tmp_path, overlap):
    out = tmp_path / 'ddp.json'
    env = dict(os.environ, BHNERF_DDP_OUT=str(out), HSA_ENABLE_IPC_MODE_LEGACY='0', BHNERF_BATCH_SEED='7', BHNERF_DDP_OVERLAP=overlap)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'ddp_worker.py')]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    r = json.load(open(out))
    assert r['world'] == 2 and r['identical'], r                      # bitwise identical parameters on both ranks
    assert len(set(r['picks'])) >= 1 and all(0 <= k < 3 for k in r['picks'])
    assert len(r['loss_vector']) == 2                                  # every rank's per-device chi^2 sum (network.py:620)
    # DP == one process on the whole batch with grad / world, up to f32 summation order: Adam turns a gradient that is
    # zero up to rounding into a +-lr step, so a handful of parameters may differ by up to the movement itself
    assert r['moved'] > 0 and r['frac_off'] < 2e-3 and r['max_diff'] <= 2.5 * r['moved'], r


@pytest.mark.timeout(600)
@pytest.mark.parametrize('overlap', ['0', '1'])
def test_rccl_process_group_of_one_rank(tmp_path, overlap):
    """The shipped multi-GPU design is one process per GPU over RCCL (backend 'nccl'); a 1-GPU box can run it at world
    size 1: init_process_group('nccl', device_id=...), the flat all-reduce on the device (synchronous, and the async work
    handle + stream wait of overlap_allreduce), then Adam -- bitwise equal to the step without a process group."""
    out = tmp_path / 'ddp.json'
    env = dict(os.environ, BHNERF_DDP_OUT=str(out), HSA_ENABLE_IPC_MODE_LEGACY='0', BHNERF_BATCH_SEED='7', BHNERF_DDP_OVERLAP=overlap,
               BHNERF_DDP_BACKEND='nccl')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'ddp_worker.py')]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    r = json.load(open(out))
    assert r['world'] == 1 and r['backend'] == 'nccl' and r['identical'], r
    assert len(r['loss_vector']) == 1 and r['moved'] > 0
    assert r['bitwise_equal_single'] and r['max_diff'] == 0.0, r


@pytest.mark.timeout(600)
def test_bench_forced_process_group_runs_rccl_on_one_gpu():
    """bench.py --gpus 1 with BHNERF_BENCH_FORCE_DIST=1 goes through init_process_group('nccl') + the flat all-reduce."""
    env = dict(os.environ, BHNERF_BENCH_FORCE_DIST='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--image', '32',
           '--ngeo', '32', '--frames', '8', '--frames-per-gpu', '2', '--width', '64', '--no-cpu-baseline', '--no-parity-mode',
           '--no-tutorial-domain']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    r = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert r['n_gpus'] == 1 and 'nccl' in r['config']['parallelism'] and r['value'] > 0


@pytest.mark.timeout(900)
def test_bench_gpus_flag_spawns_the_ranks():
    env = dict(os.environ, BHNERF_BENCH_ONE_DEVICE='1')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--image', '32',
           '--ngeo', '32', '--frames', '8', '--frames-per-gpu', '2', '--width', '64', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1
    r = json.loads(line[0])
    assert r['n_gpus'] == 2 and r['config']['frames_per_step'] == 4 and r['config']['parallelism'].startswith('dp2')
    assert r['value'] > 0 and r['scaling'] == 'weak'
    # --gpus and WORLD_SIZE must agree
    bad = subprocess.run(cmd[:2] + ['--gpus', '2', '--no-cpu-baseline'], env=dict(env, WORLD_SIZE='1', RANK='0'), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE' in (bad.stderr + bad.stdout)


@pytest.mark.timeout(1500)
def test_bench_with_eight_ranks_on_one_device():
    """The driver's 8-GPU launch shape, `bench.py --gpus 8` (8 ranks x 8 frames = 64 frames per step, one all-reduce), with every
    rank on cuda:0 over gloo (BHNERF_BENCH_ONE_DEVICE=1): rank counts beyond 1 / 2 / 4 have no other coverage on a 1-GPU box."""
    env = dict(os.environ, BHNERF_BENCH_ONE_DEVICE='1')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--image', '32',
           '--ngeo', '32', '--frames', '64', '--frames-per-gpu', '8', '--width', '64', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1400)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1
    r = json.loads(line[0])
    assert r['n_gpus'] == 8 and r['config']['frames_per_step'] == 64 and r['config']['parallelism'].startswith('dp8')
    assert r['value'] > 0 and r['scaling'] == 'weak'
