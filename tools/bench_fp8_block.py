"""FP8 block-wise (DeepSeek-V3 layout) kernels on Mixtral-8x7B expert Linear shapes: rates of act_quant, weight_cast_to_fp8 and the
block-scaled fp8 GEMM (llmc_fp8_act_quant / llmc_fp8_block_quant / llmc_fp8_block_gemm), HIP events, median of 5."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llmc_amd.compression.quantization import kernel as KN


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    M = 16384
    shapes = ((14336, 4096), (4096, 14336)) + (((7168, 7168),) if '--more' in sys.argv else ())
    for N, K in shapes:
        x = (torch.randn(M, K, generator=g, device='cuda') * torch.exp(0.5 * torch.randn(K, generator=g, device='cuda'))).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).to(torch.bfloat16)
        t_a = timed(lambda: KN.act_quant(x, 128))
        t_w = timed(lambda: KN.weight_cast_to_fp8(w, 128))
        a8, a_s = KN.act_quant(x, 128)
        w8, w_s = KN.weight_cast_to_fp8(w, 128)
        t_d = timed(lambda: KN.weight_cast_to_bf16(w8, w_s, 128))
        t_g = timed(lambda: KN.fp8_gemm(a8, a_s, w8, w_s))
        t_f = timed(lambda: KN.fp8_gemm(a8, a_s, w8, w_s, fused_scale=True))
        ref, fus = KN.fp8_gemm(a8, a_s, w8, w_s, dtype=torch.float32), KN.fp8_gemm(a8, a_s, w8, w_s, dtype=torch.float32, fused_scale=True)
        dev = float((ref - fus).abs().max() / ref.abs().max())
        fl = 2.0 * M * N * K
        print(f'M={M} N={N} K={K}: act_quant {t_a*1e6:.0f} us = {3.0*M*K/t_a/1e12:.2f} TB/s (2MK read + MK write) | '
              f'weight_cast_to_fp8 {t_w*1e6:.0f} us = {3.0*N*K/t_w/1e12:.2f} TB/s | weight_cast_to_bf16 {t_d*1e6:.0f} us = {3.0*N*K/t_d/1e12:.2f} TB/s | fp8_gemm {t_g*1e3:.2f} ms = '
              f'{fl/t_g/1e12:.0f} TFLOP/s = {fl/t_g/5e15:.3f} of the 5 PF fp8 MFMA peak | fused_scale=True {t_f*1e3:.2f} ms = {fl/t_f/1e12:.0f} TFLOP/s = '
              f'{fl/t_f/5e15:.3f} (max |difference| to the bit-identical form / max |output|, fp32 output: {dev:.1e})', flush=True)


if __name__ == '__main__':
    main()
