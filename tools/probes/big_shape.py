"""Llama-3-70B down_proj-sized run (K = 28672): Hessian -> prep -> factor -> column loop, checks U H U^T = I."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd.compression.quantization import gptq_ops
from llmc_amd.compression.quantization.hessian import HessianAccumulator

K, R, T = int(os.environ.get('K', 28672)), 1024, 32768
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(T, K, generator=g, device='cuda').to(torch.bfloat16)
W = (torch.randn(R, K, generator=g, device='cuda') * 0.02).to(torch.bfloat16)
acc = HessianAccumulator(K, 'cuda')
torch.cuda.synchronize(); t0 = time.time()
acc.add(x.unsqueeze(0))
perm = torch.argsort(torch.diagonal(acc.H), descending=True)
Hp, Wp = gptq_ops.hessian_prep(acc.H, W, perm, 0.01)
Hkeep = Hp.clone()
U = gptq_ops.chol_inv_upper(Hp, check=True)
tmp, losses, s, z = gptq_ops.gptq_quantize(Wp, U, False, 0.0, 15.0, 128)
torch.cuda.synchronize(); t1 = time.time()
# spot-check the factor on a 2048-wide trailing slab: (U H U^T)[-2048:, -2048:] = I
n = 2048
Ut = U[-n:, :].double()
E = Ut @ Hkeep.double() @ Ut.T - torch.eye(n, device='cuda', dtype=torch.float64)
print(f'K={K} R={R} T={T}: {t1 - t0:.2f} s, max|U H U^T - I| (last {n} rows) = {E.abs().max().item():.2e}, '
      f'loss sum {losses.sum().item():.4f}, finite={bool(torch.isfinite(tmp).all())}', flush=True)
