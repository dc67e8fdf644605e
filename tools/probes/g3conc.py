"""gemm3 on a side stream while the main stream runs unrelated small GEMMs: does its result change?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi

L = _ffi.lib()
torch.manual_seed(0)
M = N = 3072
Kd = 512
A = torch.randn(Kd, 4096, device='cuda')
C0 = torch.randn(M, N, device='cuda')
a2 = torch.randn(128, 4096, device='cuda')
c2 = torch.randn(512, 4096, device='cuda')
side = torch.cuda.Stream()


def g3(C, st):
    _ffi.check(L.llmc_test_gemm3(A.data_ptr(), A.data_ptr(), C.data_ptr(), A.stride(0), A.stride(0), C.stride(0), M, N,
                                 Kd, 1, st), 'g3')


ref = C0.clone()
g3(ref, _ffi.stream())
torch.cuda.synchronize()
for trial in range(6):
    C = C0.clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        g3(C, side.cuda_stream)
    for _ in range(40):   # unrelated short-K products on the current stream
        _ffi.check(L.llmc_test_sgemm(a2.data_ptr(), a2.data_ptr(), c2.data_ptr(), a2.stride(0), a2.stride(0),
                                     c2.stride(0), 512, 4096, 128, 1, 0, 0, 0, 0, 0, 1, _ffi.stream()), 'sg')
    torch.cuda.synchronize()
    print(trial, int((C != ref).sum().item()), flush=True)
