"""Host issue time vs total time of llmc_chol_inv_upper / llmc_gptq_quantize under the schedules of round 4.
usage: python tools/probes/k3_time.py [K] [R]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi
from llmc_amd.compression.quantization import gptq_ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = torch.Generator(device='cuda').manual_seed(0)
X = torch.randn(2 * K, K, generator=g, device='cuda')
H = (X.T @ X) / K
H += 0.01 * torch.diagonal(H).mean() * torch.eye(K, device='cuda')
del X
W = torch.randn(R, K, generator=g, device='cuda') * 0.02


def run(label, fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[1]:
            best = (t1 - t0, t2 - t0)
    print(f'{label:64s} host issue {best[0] * 1e3:7.2f} ms   total {best[1] * 1e3:7.2f} ms', flush=True)
    return out


side = torch.cuda.Stream()
for where in ('default stream', 'side stream'):
    ctx = torch.cuda.stream(side) if where == 'side stream' else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        os.environ.pop('LLMC_K3_PIPE', None)
        U = run(f'K3 K={K} pipelined, caller on the {where}', lambda: gptq_ops.chol_inv_upper(H.clone(), check=False))
        os.environ['LLMC_K3_PIPE'] = '0'
        run(f'K3 K={K} round-3 schedule (one helper stream), {where}', lambda: gptq_ops.chol_inv_upper(H.clone(), check=False))
        with _ffi.helper_streams(False):
            run(f'K3 K={K} single stream, {where}', lambda: gptq_ops.chol_inv_upper(H.clone(), check=False))
        os.environ.pop('LLMC_K3_PIPE', None)
        run(f'K4 R={R} K={K} pipelined (bulk stream), {where}', lambda: gptq_ops.gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128))
        with _ffi.helper_streams(False):
            run(f'K4 R={R} K={K} single stream, {where}', lambda: gptq_ops.gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128))
