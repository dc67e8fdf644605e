"""FP8 (e4m3fn) block-wise ops with the names and signatures of llmc/compression/quantization/kernel.py (the reference's
Triton kernels, gated there on a Hopper check, utils.py:23-28) — here HIP kernels for gfx950 (fp8_block.hip)."""
import torch

from llmc_amd import _ffi


def act_quant(x, block_size=128):
    """kernel.py:31-55: per `block_size` consecutive elements of the last dim: s = absmax / 448, y = e4m3(x / s).
    Returns (y float8_e4m3fn like x, s fp32 [..., K / block_size])."""
    _ffi.require_gpu(x)
    assert x.is_contiguous(), 'Input tensor must be contiguous'
    assert x.size(-1) % block_size == 0, f'Last dimension size must be divisible by block_size (block_size={block_size})'
    L = _ffi.lib()
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    s = torch.empty(*x.shape[:-1], x.size(-1) // block_size, dtype=torch.float32, device=x.device)
    _ffi.check(L.llmc_fp8_act_quant(_ffi.ptr(x), _ffi.dt(x), x.numel(), int(block_size), _ffi.ptr(y), _ffi.ptr(s),
                                    _ffi.stream()), 'llmc_fp8_act_quant')
    return y.view(torch.float8_e4m3fn), s


def weight_cast_to_fp8(x, block_size=128, clamp_min=0.0):
    """kernel.py:75-86: 128 x 128 blocks of a 2-D weight -> (float8_e4m3fn [M, N], fp32 scales [ceil(M/b), ceil(N/b)])."""
    _ffi.require_gpu(x)
    assert x.is_contiguous() and x.dim() == 2
    L = _ffi.lib()
    M, N = x.shape
    y = torch.empty((M, N), dtype=torch.uint8, device=x.device)
    s = torch.empty((-(-M // block_size), -(-N // block_size)), dtype=torch.float32, device=x.device)
    _ffi.check(L.llmc_fp8_block_quant(_ffi.ptr(x), _ffi.dt(x), M, N, int(block_size), float(clamp_min), 0, _ffi.ptr(y),
                                      _ffi.ptr(s), _ffi.stream()), 'llmc_fp8_block_quant')
    return y.view(torch.float8_e4m3fn), s


def weight_cast_to_bf16(x, s, block_size=128, dtype=torch.bfloat16):
    """kernel.py:112-143 / quant.py:18-30: dequantize a block-scaled e4m3 weight."""
    _ffi.require_gpu(x, s)
    assert x.is_contiguous() and s.is_contiguous(), 'Input tensors must be contiguous'
    assert x.dim() == 2 and s.dim() == 2, 'Input tensors must have 2 dimensions'
    L = _ffi.lib()
    M, N = x.shape
    y = torch.empty((M, N), dtype=dtype, device=x.device)
    _ffi.check(L.llmc_fp8_block_dequant(_ffi.ptr(x.view(torch.uint8)), _ffi.ptr(s.float().contiguous()), M, N,
                                        int(block_size), _ffi.dt(dtype), _ffi.ptr(y), _ffi.stream()),
               'llmc_fp8_block_dequant')
    return y


def fp8_gemm(a, a_s, b, b_s, dtype=torch.bfloat16, bias=None, fused_scale=False):
    """kernel.py:213-242: c[..., N] = sum over 128-deep K blocks (a_blk . b_blk^T) * a_s * b_s.
    fused_scale (not a reference argument; default False = bit-identical to the reference's Triton kernel): the K-block update as
    ONE fma with the scale product rounded once (LLMC_FP8_GEMM_FUSED_SCALE) — faster, within an fp32 rounding per K block."""
    _ffi.require_gpu(a, a_s, b, b_s, bias)
    assert a.is_contiguous() and b.is_contiguous(), 'Input tensors must be contiguous'
    assert a_s.is_contiguous() and b_s.is_contiguous(), 'Scaling factor tensors must be contiguous'
    L = _ffi.lib()
    K = a.size(-1)
    M = a.numel() // K
    N = b.size(0)
    c = torch.empty(*a.shape[:-1], N, dtype=dtype, device=a.device)
    if bias is not None:
        bias = bias.to(dtype).contiguous()
    _ffi.check(L.llmc_fp8_block_gemm(_ffi.ptr(a.view(torch.uint8)), _ffi.ptr(a_s), _ffi.ptr(b.view(torch.uint8)),
                                     _ffi.ptr(b_s), M, N, K, _ffi.dt(dtype) | (0x100 if fused_scale else 0), _ffi.ptr(bias), _ffi.ptr(c), _ffi.stream()),
               'llmc_fp8_block_gemm')
    return c


def block_wise_fp8_forward_func(x, w, w_scale, block_size, bias):
    """module_utils.py:40-45: quantize the activation per 1 x block, multiply by the block-scaled fp8 weight."""
    if block_size != 128:
        raise NotImplementedError('block_wise_fp8_forward_func: the GEMM kernel takes 128-deep blocks')
    xq, scale = act_quant(x.contiguous(), block_size)
    return fp8_gemm(xq, scale, w, w_scale.contiguous(), dtype=torch.bfloat16, bias=bias)
