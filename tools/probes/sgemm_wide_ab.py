"""A/B of K4's phased far update (C -= Err^T Hinv rows, phase 128): k_sgemm_wide (256 x 128 tiles, LDS-DMA, one wave per SIMD)
against k_sgemm (128 x 128, two workgroups per CU) on the column loop's shapes; same bits (checked). One MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llmc_amd import _ffi


COLD = None


def run(tag, M, N, Kd, K=None, reps=7):
    L = _ffi.lib()
    K = K or N
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(Kd, M, device='cuda', generator=g) * 0.01
    Bfull = torch.randn(Kd, K, device='cuda', generator=g)
    C0 = torch.randn(M, K, device='cuda', generator=g)
    B = Bfull[:, K - N:]
    out, t = {}, {}
    for arm, opts in (('wide', dict(sgemm_no_wide=4)), ('wide2', dict(sgemm_no_wide=2)), ('k_sgemm', dict(sgemm_no_wide=1))):
        with _ffi.option(**opts):
            C = C0.clone()
            Cv = C[:, K - N:]
            def go():
                _ffi.check(L.llmc_test_sgemm_phased(A.data_ptr(), B.data_ptr(), Cv.data_ptr(), A.stride(0), B.stride(0), C.stride(0),
                                                    M, N, Kd, 1, 128, _ffi.stream()), 'sgemm_phased')
            go()
            torch.cuda.synchronize()
            out[arm] = C.clone()
            ts = []
            for _ in range(reps):
                if COLD is not None:
                    COLD.add_(1.0)          # 1 GiB read + written: nothing of the operands left in L2 / MALL
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); go(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            t[arm] = sorted(ts)[len(ts) // 2]
    same = all(torch.equal(out[k].view(torch.int32), out['k_sgemm'].view(torch.int32)) for k in ('wide', 'wide2'))
    fl = 2.0 * M * N * Kd
    print(f'{tag:22s} M={M:6d} N={N:6d} Kd={Kd}: 256x128 {t["wide"]:8.1f} us ({fl/t["wide"]/1e6/157.3:.3f} of fp32 MFMA peak) | '
          f'128x128 x2 {t["wide2"]:8.1f} us ({fl/t["wide2"]/1e6/157.3:.3f}) | '
          f'k_sgemm {t["k_sgemm"]:8.1f} us ({fl/t["k_sgemm"]/1e6/157.3:.3f}) | same bits: {same}', flush=True)


if __name__ == '__main__':
    if '--cold' in sys.argv:
        COLD = torch.zeros(1 << 28, device='cuda')
    if '--ld' in sys.argv:          # one round of tiles (4096 x 2048), the operands' row stride varied
        for K in (2048, 4096, 6144, 8192, 12288, 14336, 14336 + 64, 16384, 28672):
            run(f'ld sweep K={K}', 4096, 2048, 512, K)
        for M in (2048, 4096, 8192, 16384):
            run(f'M sweep M={M}', M, 4096, 512, 14336)
        sys.exit(0)
    run('down 8B first group', 4096, 13824, 512, 14336)
    run('down 8B mid', 4096, 7168, 512, 14336)
    run('down 8B late', 4096, 2048, 512, 14336)
    run('down 8B last', 4096, 512, 512, 14336)
    run('gate|up 8B first', 28672, 3584, 512, 4096)
    run('gate|up 8B mid', 28672, 2048, 512, 4096)
    run('q|k|v 8B first', 6144, 3584, 512, 4096)
    run('o 8B first', 4096, 3584, 512, 4096)
    run('o 8B mid', 4096, 1536, 512, 4096)
    run('down 70B first', 8192, 28160, 512, 28672)
    run('gate|up 70B first', 57344, 7680, 512, 8192)
