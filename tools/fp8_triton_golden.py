"""Goldens for the reference's TRITON-only FP8 block-wise ops (llmc/compression/quantization/kernel.py:7-242: act_quant,
weight_cast_to_fp8, weight_cast_to_bf16, fp8_gemm), produced by running the UNMODIFIED reference kernels on the MI355X
(Triton on ROCm; oracle/_ref_gpu is a plain copy of the reference made by oracle/build_ref.py). They cannot run in the
build container (no GPU), so this script runs on the GPU box:

    python tools/fp8_triton_golden.py gpurun_out/fp8_triton.npz

and the result is committed as tests/golden/fp8_triton.npz. Inputs are stored with the outputs. Test infrastructure."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_kernel_module():
    p = os.path.join(ROOT, 'oracle', '_ref_gpu', 'llmc', 'compression', 'quantization', 'kernel.py')
    spec = importlib.util.spec_from_file_location('ref_kernel', p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/fp8_triton.npz'
    K = load_kernel_module()
    dev = torch.device('cuda', 0)
    g = {}
    # act_quant: bf16 and f16 inputs, an all-zero block (scale 0 -> 0/0)
    for dt, name in ((torch.bfloat16, 'bf16'), (torch.float16, 'f16')):
        gen = torch.Generator().manual_seed(3)
        x = (torch.randn(5, 37, 512, generator=gen) * torch.exp(torch.randn(512, generator=gen))).to(dt)
        x[0, 0, :128] = 0.0
        y, s = K.act_quant(x.to(dev).contiguous(), 128)
        g[f'aq_{name}_x'] = x.float().numpy()
        g[f'aq_{name}_bits'] = y.view(torch.uint8).cpu().numpy()
        g[f'aq_{name}_scales'] = s.cpu().numpy()
    # weight casts + GEMM
    shapes = [(256, 256, 512), (64, 1024, 1024), (300, 200, 384)]
    g['n_gemm'] = np.int64(len(shapes))
    for i, (M, N, Kd) in enumerate(shapes):
        gen = torch.Generator().manual_seed(M + N)
        x = (torch.randn(M, Kd, generator=gen) * torch.exp(0.5 * torch.randn(Kd, generator=gen))).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, Kd, generator=gen) * 0.05).to(torch.bfloat16).to(dev)
        a8, a_s = K.act_quant(x.contiguous(), 128)
        w8, w_s = K.weight_cast_to_fp8(w.contiguous(), 128)
        back = K.weight_cast_to_bf16(w8, w_s, 128)
        torch.set_default_dtype(torch.bfloat16)          # the DeepSeek flow the op serves runs with a bf16 default dtype
        c = K.fp8_gemm(a8, a_s, w8, w_s)
        torch.set_default_dtype(torch.float32)
        c32 = K.fp8_gemm(a8, a_s, w8, w_s)
        p = f'g{i}_'
        g[p + 'shape'] = np.array([M, N, Kd])
        g[p + 'x'] = x.float().cpu().numpy()
        g[p + 'w'] = w.float().cpu().numpy()
        g[p + 'a_bits'] = a8.view(torch.uint8).cpu().numpy()
        g[p + 'a_s'] = a_s.cpu().numpy()
        g[p + 'w_bits'] = w8.view(torch.uint8).cpu().numpy()
        g[p + 'w_s'] = w_s.cpu().numpy()
        g[p + 'back'] = back.float().cpu().numpy()
        g[p + 'back_dtype'] = np.array(str(back.dtype))
        g[p + 'c_bf16'] = c.float().cpu().numpy()
        g[p + 'c_f32'] = c32.float().cpu().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **g)
    print('wrote', out, {k: v.shape for k, v in g.items() if hasattr(v, 'shape') and v.ndim > 0 and v.size > 8})


if __name__ == '__main__':
    main()
