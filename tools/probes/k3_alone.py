"""One chol_inv_upper (K3) and one column loop (K4) of the bench's down_proj shape alone on the device, for a rocprofv3 kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d out -o ks -- python tools/probes/k3_alone.py [K] [R]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi
from llmc_amd.compression.quantization import gptq_ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
X = torch.randn(2 * K, K, device='cuda')
H = (X.T @ X) / K
H += 0.01 * H.diag().mean() * torch.eye(K, device='cuda')
del X
W = torch.randn(R, K, device='cuda') * 0.02
with _ffi.helper_streams(False):
    for it in range(3):
        A = H.clone()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        U = gptq_ops.chol_inv_upper(A, check=False)
        e1.record()
        gptq_ops.gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128)
        e2.record()
        torch.cuda.synchronize()
        print(f'K = {K} R = {R}: chol_inv_upper {e0.elapsed_time(e1):.2f} ms, column loop {e1.elapsed_time(e2):.2f} ms', flush=True)
