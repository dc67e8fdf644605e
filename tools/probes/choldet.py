"""Run-to-run determinism of llmc_chol_inv_upper (a race shows up as differing bits)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper

for K in (4096, 1536):
    g = torch.Generator().manual_seed(K)
    X = torch.randn(3 * K, K, generator=g, dtype=torch.float64)
    X[:, ::7] *= 5
    H = (X.T @ X) * (2.0 / 3)
    H += 0.01 * H.diag().mean() * torch.eye(K, dtype=torch.float64)
    Hd = H.float().cuda()
    outs = [chol_inv_upper(Hd.clone()).clone() for _ in range(8)]
    torch.cuda.synchronize()
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    errs = [float((o.double().cpu() - ref).abs().max() / ref.abs().max()) for o in outs]
    print(K, 'diff vs run0:', [int((outs[0] != o).sum().item()) for o in outs[1:]], flush=True)
    print(K, 'rel err vs fp64:', ['%.2e' % e for e in errs], flush=True)
