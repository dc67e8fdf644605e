"""down_proj's K3 + K4 once (K = 14336, R = 4096), for a rocprofv3 --pmc pass: which of the chain's product kernels keep the
MFMA pipe busy, and at what clock (tools/probes/pmc_table.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmc_amd.compression.quantization import gptq_ops

K, R, T = 14336, 4096, 16384
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(T, K, generator=g, device='cuda')
H = (x.T @ x) / T
H += 0.01 * torch.diagonal(H).mean() * torch.eye(K, device='cuda')
del x
W = (torch.randn(R, K, generator=g, device='cuda') * 0.02).to(torch.bfloat16)
for _ in range(2):
    U = gptq_ops.chol_inv_upper(H.clone(), check=False)
    out = gptq_ops.gptq_quantize(W, U, False, 0.0, 15.0, 128)
    torch.cuda.synchronize()
