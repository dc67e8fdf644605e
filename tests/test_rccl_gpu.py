"""The cooperative multi-GPU modes over RCCL on real hardware (skipped on a box with fewer than 2 GPUs): a 2-rank
block — activations or the Hessian broadcast over xGMI, layers dealt round-robin, tensors gathered point-to-point —
gives the bits of the single-GPU run; the sample-sharded subset (one all_reduce, row-sharded loop) agrees statistically."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs of one node')
def test_two_rank_rccl_block_equals_single_gpu():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'rccl_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
