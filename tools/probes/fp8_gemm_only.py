"""The block-scaled fp8 GEMM alone (for rocprofv3 --pmc): three launches per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd.compression.quantization import kernel as KN

g = torch.Generator(device='cuda').manual_seed(0)
M = 16384
for N, K in ((14336, 4096), (4096, 14336)):
    a8 = (torch.randn(M, K, generator=g, device='cuda') * 2).to(torch.float8_e4m3fn)
    w8 = torch.randn(N, K, generator=g, device='cuda').to(torch.float8_e4m3fn)
    a_s = torch.rand(M, K // 128, generator=g, device='cuda') + 0.5
    w_s = torch.rand(N // 128, K // 128, generator=g, device='cuda') * 0.1 + 0.01
    for _ in range(3):
        KN.fp8_gemm(a8, a_s, w8, w_s)
    torch.cuda.synchronize()
