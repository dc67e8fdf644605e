"""Micro-benchmark of llmc_hessian_accum on MI355X: TFLOP/s against the 2.5 PF dense bf16 MFMA peak.
Contract flops: F_H = T*K*(K+1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from llmc_amd.compression.quantization.hessian import HessianAccumulator


def run(T, K, dt=torch.bfloat16, reps=5):
    g = torch.Generator(device='cuda').manual_seed(1)
    fill = os.environ.get('FILL', 'randn')
    if fill == 'zeros':
        x = torch.zeros(T, K, device='cuda', dtype=dt)
    elif fill == 'ones':
        x = torch.ones(T, K, device='cuda', dtype=dt)
    else:
        x = torch.randn(T, K, generator=g, device='cuda', dtype=torch.float32).to(dt)
    acc = HessianAccumulator(K, 'cuda')
    acc.add(x.unsqueeze(0))
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        acc.nsamples = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        acc.add(x.unsqueeze(0))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    fl = T * K * (K + 1)
    print(f'T={T} K={K} {dt}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s contract '
          f'({fl/t/2.5e15*100:.1f}% of 2.5 PF), GEMM-equiv {2*T*K*K/t/1e12:.1f}', flush=True)


if __name__ == '__main__':
    shapes = [(2048, 4096), (32768, 4096), (262144, 4096), (65536, 14336)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    for T, K in shapes:
        run(T, K)
