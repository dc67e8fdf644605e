"""One K3 factorisation + one K4 column loop per shape, for a rocprofv3 --kernel-trace timeline (tools/r02_gpu_e.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd.compression.quantization import gptq_ops

shapes = [(14336, 4096), (4096, 28672)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for K, R in shapes:
    g = torch.Generator(device='cuda').manual_seed(0)
    X = torch.randn(2 * K, K, generator=g, device='cuda')
    H = (X.T @ X) / K
    H += 0.01 * torch.diagonal(H).mean() * torch.eye(K, device='cuda')
    del X
    W = torch.randn(R, K, generator=g, device='cuda') * 0.02
    for it in range(2):
        U = gptq_ops.chol_inv_upper(H.clone(), check=False)
        torch.cuda.synchronize()
        tmp, losses, s, z = gptq_ops.gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128)
        torch.cuda.synchronize()
    print('done', K, R, flush=True)
