"""More goldens of the reference's Triton fp8_gemm (llmc/compression/quantization/kernel.py:146-242), run UNMODIFIED on the MI355X
(oracle/_ref_gpu, see tools/fp8_triton_golden.py): more shapes — several 128-column blocks of b_s, up to 16 K blocks, M below
one tile, ragged M / N — and the fp32 output (it shows the accumulation order without the bf16 rounding on top; the bf16 output is checked to be
its rounding).
Inputs are stored as the fp8 codes and fp32 scales the GEMM takes (act_quant / weight_cast_to_fp8 are pinned by
fp8_triton.npz). Runs on the GPU box:   python tools/fp8_triton_golden_more.py gpurun_out/fp8_triton_more.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fp8_triton_golden import load_kernel_module  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/fp8_triton_more.npz'
    K = load_kernel_module()
    dev = torch.device('cuda', 0)
    g = {}
    # small enough to commit (codes do not compress): 8 K blocks over two b_s column blocks; M below one tile (the K = 16 MFMA
    # kernel); one K block with ragged M / N; five K blocks, three column blocks; twelve K blocks, ragged both ways
    shapes = [(192, 256, 1024), (64, 256, 512), (130, 136, 128), (320, 384, 640), (300, 200, 1536)]
    g['n_gemm'] = np.int64(len(shapes))
    for i, (M, N, Kd) in enumerate(shapes):
        gen = torch.Generator().manual_seed(7 * M + N + Kd)
        x = (torch.randn(M, Kd, generator=gen) * torch.exp(0.5 * torch.randn(Kd, generator=gen))).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, Kd, generator=gen) * 0.05 * torch.exp(0.3 * torch.randn(N, 1, generator=gen))).to(torch.bfloat16).to(dev)
        a8, a_s = K.act_quant(x.contiguous(), 128)
        w8, w_s = K.weight_cast_to_fp8(w.contiguous(), 128)
        torch.set_default_dtype(torch.bfloat16)
        c = K.fp8_gemm(a8, a_s, w8, w_s)
        torch.set_default_dtype(torch.float32)
        c32 = K.fp8_gemm(a8, a_s, w8, w_s)
        assert c.dtype == torch.bfloat16 and c32.dtype == torch.float32
        p = f'g{i}_'
        g[p + 'shape'] = np.array([M, N, Kd])
        g[p + 'a_bits'] = a8.view(torch.uint8).cpu().numpy()
        g[p + 'a_s'] = a_s.cpu().numpy()
        g[p + 'w_bits'] = w8.view(torch.uint8).cpu().numpy()
        g[p + 'w_s'] = w_s.cpu().numpy()
        assert torch.equal(c, c32.to(torch.bfloat16))      # the bf16 output is the fp32 accumulator rounded once: store one
        g[p + 'c_f32'] = c32.cpu().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **g)
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')


if __name__ == '__main__':
    main()
