"""A/B of K3 far-update kernels (round 6): k_gemm3w (two 128 x 128 workgroups per CU) against k_gemm3s (option gemm3_no_wide), on the far
update of a K = 14336 factorisation's first outer blocks (C -= P^T P, upper only, Kd = 512) and on whole factorisations.
usage: python tools/probes/gemm3w_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi
from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper

L = _ffi.lib()
Kd = 512
for n in (13312, 8192, 3584):
    P = torch.randn(Kd, n, device='cuda')
    C0 = torch.randn(n, n, device='cuda')
    ws = torch.empty(6 * Kd * n, dtype=torch.int16, device='cuda')
    outs = {}
    for no_dma in (1, 0, 1, 0):   # 1 = k_gemm3s, 0 = k_gemm3w
        with _ffi.option(gemm3_no_wide=no_dma):
            ts = []
            for it in range(8):
                C = C0.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _ffi.check(L.llmc_test_gemm3_planes(P.data_ptr(), P.data_ptr(), C.data_ptr(), n, n, n, n, n, Kd, 0, 1, ws.data_ptr(), _ffi.stream()), 'planes')
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            outs[no_dma] = C
            ts = sorted(ts[3:])
            print(f'n = {n:6d}  split + {"k_gemm3s" if no_dma else "k_gemm3w"}: {ts[len(ts) // 2]:8.1f} us', flush=True)
    iu = torch.triu(torch.ones(n, n, dtype=torch.bool, device='cuda'))
    print('   same bits:', bool(torch.equal(outs[0][iu], outs[1][iu])), flush=True)

for K in (14336, 4096, 8192):
    X = torch.randn(2 * K, K, device='cuda')
    H = (X.T @ X) / K
    H += 0.01 * H.diag().mean() * torch.eye(K, device='cuda')
    del X
    res = {}
    for no_dma in (1, 0, 1, 0):   # 1 = k_gemm3s, 0 = k_gemm3w
        with _ffi.option(gemm3_no_wide=no_dma), _ffi.helper_streams(False):
            ts = []
            for it in range(5):
                A = H.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                U = chol_inv_upper(A, check=False)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[no_dma] = U.clone()
            print(f'K = {K:6d}  chol_inv_upper (one stream), {"k_gemm3s" if no_dma else "k_gemm3w"}: {sorted(ts[2:])[1]:8.2f} ms', flush=True)
    print('   same bits:', bool(torch.equal(res[0], res[1])), flush=True)
