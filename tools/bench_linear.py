"""F.linear on the HIP GEMMs: row-major 8-wave kernel vs k-tiled one-wave-per-SIMD kernel (incl. packing x per call)."""
import torch

from llmc_amd.compression.quantization import awq_ops


def timed(fn, n=30, warm=12):
    # sustained rate: the clock under a long MFMA load differs from the first milliseconds after idle, so the same call warms
    # the chip for tens of milliseconds before the timed window
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, K, R) in [(65536, 4096, 4096), (65536, 4096, 14336), (65536, 14336, 4096), (16384, 4096, 6144), (2048, 4096, 4096)]:
    x = torch.randn(N, K, device='cuda', dtype=torch.bfloat16)
    w = (torch.randn(R, K, device='cuda', dtype=torch.float32) * 0.02).to(torch.bfloat16)
    b = torch.randn(R, device='cuda', dtype=torch.bfloat16)
    fl = 2.0 * N * K * R
    t_row = timed(lambda: awq_ops.linear_out(x, w, b))
    wt = awq_ops.ktile_pack(w)
    t_kt = timed(lambda: awq_ops.linear_out(awq_ops.ktile_pack(x), wt, b, tiled=True))
    t_kt_only = timed(lambda xt=awq_ops.ktile_pack(x): awq_ops.linear_out(xt, wt, b, tiled=True))
    t_blk = timed(lambda xt=awq_ops.ktile_pack(x): awq_ops.linear_out(xt, wt, b, tiled=True, blocked=True))
    t_blas = timed(lambda: torch.nn.functional.linear(x, w, b))
    print(f'N={N} K={K} R={R}: row-major {t_row:.3f} ms ({fl / t_row / 1e9:.0f} TF) | k-tiled incl. pack(x) {t_kt:.3f} ms '
          f'({fl / t_kt / 1e9:.0f} TF), GEMM alone {t_kt_only:.3f} ms ({fl / t_kt_only / 1e9:.0f} TF), tile-blocked output {t_blk:.3f} ms ({fl / t_blk / 1e9:.0f} TF) | '
          f'torch F.linear {t_blas:.3f} ms ({fl / t_blas / 1e9:.0f} TF)')
