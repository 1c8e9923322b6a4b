"""GPU: the data-parallel path end to end (rollout sharding by trajectory, global advantage statistics through the
d4_learn all-reduce callback, one flat gradient all-reduce per head, clipped AdamW) equals a single process over the
global batch.  Two ranks share the one MI355X of the test box over gloo; the 8-GPU RCCL run is the driver's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_equal_one_process_at_the_global_batch():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29671', os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('DPGPU_OK_') == 2, out.stdout


def test_eight_ranks_with_empty_shards_equal_one_process_at_the_global_batch():
    """The rank count of the 8-GPU run on the test box's one device (gloo): 6 trajectories over 8 ranks -> shard sizes 1, 1, 1, 1, 1, 1, 0, 0.  The two
    ranks with an EMPTY shard take part in the three statistics all-reduces of d4_learn and in both gradient all-reduces with zero contributions (no
    deadlock, no division by a zero local count), every rank reports the global losses and ends with identical weights, equal to one process over the
    whole batch (trainers.py:1388-1396, 1436-1452)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
           '--master-port', '29679', os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('DPGPU_OK_') == 8, out.stdout


def test_two_ranks_equal_one_process_at_the_headline_size():
    """The same at BASELINE config 3's per-rank shape: config 2's architecture (dim 512, depth 6), a global batch of 256 trajectories x 16 frames sharded
    over two ranks (128 each: every GEMM runs at other row counts than the single-process 256), a terminal bias that ends a share of the trajectories
    early so that the global advantage statistics and masked-mean denominators matter: losses, the all-reduced gradients of both heads and the weights
    after clip + AdamW equal a single process over the global batch."""
    env = dict(os.environ, D4_DP_SIZE='headline', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29675', os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('DPGPU_OK_') == 2, out.stdout


def test_two_ranks_over_rccl_equal_one_process_at_the_global_batch():
    """The same check with one MI355X per rank and backend "nccl" (= RCCL over xGMI): the gradient buckets, the global statistics and the
    parameter checks cross devices through the collective library the 8-GPU run uses (trainers.py:1388-1396, 1436-1452 — the reference's
    implicit DDP all-reduce).  Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs (one per rank for RCCL)')
    env = dict(os.environ, D4_DP_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29673', os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('DPGPU_OK_') == 2, out.stdout


@pytest.mark.parametrize('launch', ['torch.distributed.run', 'bare', 'bare8'])
def test_bench_runs_its_multi_rank_path_with_two_ranks(launch):
    """bench.py launched as the driver launches it for N = 2 (torch.distributed.run, one rank per process): strict tile table (no rank times GEMM
    tiles on its own clock), barrier + max-over-ranks timing, per-rank skew, value = the units ALL ranks processed / that time.  Two ranks share
    the test box's one GPU over gloo (D4_BENCH_BACKEND; RCCL wants a device per rank) - the collective calls and the control flow are the N-rank ones.
    'bare': `python bench.py --gpus 2` with no launcher and no WORLD_SIZE — bench.py re-executes itself under torch.distributed.run (one process per
    GPU) and still prints exactly one JSON line.  'bare8': the same with EIGHT ranks, the 8-GPU lease's shape rehearsed on one device: eight engines,
    eight strict tile-table loads, the rank-0-only line with n_gpus 8, global batch 2048, dp8."""
    import json
    n = 8 if launch == 'bare8' else 2
    env = dict(os.environ, D4_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '1']
    if launch.startswith('bare'):
        cmd = [sys.executable, *tail]
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port', '29677', *tail]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1, out.stdout                                # rank 0 prints ONE JSON line
    d = json.loads(line[0])
    assert d['n_gpus'] == n and d['config']['global_batch'] == 256 * n and d['scaling'] == 'weak' and d['config']['parallelism'] == f'dp{n}'
    assert abs(d['value'] - n * 256 * 16 / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']
    assert d['per_rank']['generate_ms_max'] >= d['per_rank']['generate_ms_min'] > 0 and 'cpu_baseline' not in d and 'cfg5_bf16' not in d
