"""Micro-benchmark of the internal fp32-MFMA GEMM (llmc_test_sgemm) on the shapes K3/K4 use."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llmc_amd import _ffi


def run(M, N, Kd, TA, TB, epi, hints=(0, 0, 0, 0), reps=5, tag=''):
    L = _ffi.lib()
    fill = os.environ.get('FILL', 'randn')   # zeros: how much of the gap to peak is power (DVFS), not the kernel
    mk = torch.zeros if fill == 'zeros' else torch.randn
    A = mk((Kd, M) if TA else (M, Kd), device='cuda')
    B = mk((N, Kd) if TB else (Kd, N), device='cuda')
    C = torch.randn(M, N, device='cuda')
    def go():
        _ffi.check(L.llmc_test_sgemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), A.stride(0), B.stride(0), C.stride(0),
                                     M, N, Kd, int(TA), int(TB), epi, *hints, _ffi.stream()), 'sgemm')
    go()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    fl = 2.0 * M * N * Kd * (0.5 if (hints[3] or hints[0] or hints[2]) else 1.0)
    print(f'{tag:28s} M={M} N={N} Kd={Kd} TA={int(TA)} TB={int(TB)}: {t*1e6:9.1f} us  {fl/t/1e12:6.1f} TFLOP/s '
          f'({fl/t/157.3e12*100:.0f}% of fp32 MFMA peak), C traffic {2*4*M*N*(0.5 if hints[3] else 1)/t/1e12:.2f} TB/s', flush=True)


def run3(M, N, Kd, upper, reps=5, tag='', TA=True, epi=0, hints=(0, 0)):
    L = _ffi.lib()
    A = torch.randn((Kd, M) if TA else (M, Kd), device='cuda')
    B = torch.randn(Kd, N, device='cuda')
    C = torch.randn(M, N, device='cuda')
    def go():
        _ffi.check(L.llmc_test_gemm3(A.data_ptr(), B.data_ptr(), C.data_ptr(), A.stride(0), B.stride(0), C.stride(0),
                                     M, N, Kd, int(TA), epi, hints[0], hints[1], int(upper), _ffi.stream()), 'gemm3')
    go()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    fl = 2.0 * M * N * Kd * (0.5 if (upper or hints[0] or hints[1]) else 1.0)
    print(f'{tag:28s} M={M} N={N} Kd={Kd} split-bf16 {"TN" if TA else "NN"}: {t*1e6:9.1f} us  {fl/t/1e12:6.1f} fp32-equivalent TFLOP/s '
          f'({6*fl/t/2.5e15*100:.0f}% of the bf16 MFMA peak in issued products)', flush=True)


if __name__ == '__main__':
    run3(13312, 13312, 512, True, tag='chol far K=14336 grp x3')
    run3(3072, 3072, 512, True, tag='chol far K=4096 grp x3')
    run3(4096, 4096, 4096, False, tag='square TN x3')
    run3(4096, 4096, 4096, False, tag='square NN x3', TA=False, epi=1)
    run3(8192, 6144, 8192, False, tag='trtri top X x3', TA=False, epi=1, hints=(1, 0))
    run3(8192, 6144, 6144, False, tag='trtri top Y x3', TA=False, epi=2, hints=(0, 1))
    run(4096, 4096, 4096, False, False, 1, tag='square NN')
    run(4096, 4096, 4096, True, False, 1, tag='square TN')
    run(28672, 3968, 128, False, False, 0, tag='K4 trailing gate|up blk0')
    run(4096, 14208, 128, False, False, 0, tag='K4 trailing down blk0')
    run(6144, 2048, 128, False, False, 0, tag='K4 trailing qkv mid')
    run(14208, 14208, 128, True, False, 0, (0, 0, 0, 1), tag='chol trailing K=14336 blk0')
    run(128, 14208, 128, True, False, 1, (0, 1, 0, 0), tag='chol panel K=14336 blk0')
    run(8192, 6144, 8192, False, False, 1, (1, 0, 0, 0), tag='trtri top X=A^-1 C')
    run(8192, 6144, 6144, False, False, 2, (0, 0, 1, 0), tag='trtri top Y=-X B^-1')
    run(2048, 2048, 2048, False, False, 1, (1, 0, 0, 0), tag='trtri K=4096 top X=A^-1 C')
    run(4096, 9216, 512, False, False, 0, tag='K4 far down grp')
    run(28672, 3072, 512, False, False, 0, tag='K4 far gate|up grp')
    run(13312, 13312, 512, True, False, 0, (0, 0, 0, 1), tag='chol far K=14336 grp')
    run(3072, 3072, 512, True, False, 0, (0, 0, 0, 1), tag='chol far K=4096 grp')
