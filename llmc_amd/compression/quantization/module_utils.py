"""Linear wrappers with llmc's `Cls.new(module, **params)` protocol (llmc/compression/quantization/
module_utils.py:586-1065). Buffers named `buf_*` carry qparams between phases exactly like the reference
(module_utils.py:597-602). Packing runs on the GPU (the reference round-trips through numpy on the host,
module_utils.py:846-860, or loops over columns in Python, :1018-1046)."""
from functools import partial

import torch
import torch.nn as nn

from .quant import pack_awq_gemm, pack_lsb


def hip_linear(x, weight, bias):
    """y = x W^T + b for the fake-quant wrappers (module_utils.py:619-644, 706-741): the HIP MFMA GEMMs
    (awq_ops.linear_auto: the k-tiled one-wave-per-SIMD kernel whenever K % 128 == 0 — a product with fewer 256 x 256 output
    tiles than CUs, e.g. an evaluation forward of 1 x 2048 tokens through a 4096-wide layer = 128 tiles, is cut into k-slices
    so that tiles x slices fills the chip, fp32 partials summed and rounded once; the row-major kernel for K % 64 == 0; fp32
    accumulation, one rounding). Round 5: no product of a supported shape goes to the vendor GEMM any more."""
    from llmc_amd import _ffi

    from . import awq_ops
    _ffi.require_gpu(x, weight)
    if weight.dtype != x.dtype:
        weight = weight.to(x.dtype)
    return awq_ops.linear_auto(x, weight, bias)


def _func_name(f):
    return f.func.__name__ if isinstance(f, partial) else f.__name__


def _copy_buf(dst, src):
    for name, buf in src.named_buffers():
        if name.startswith('buf_'):
            dst.register_buffer(name, buf.data)
    for name, p in src.named_parameters():
        if name.startswith('buf_'):
            dst.register_buffer(name, p.data)


class OriginFloatLinear(nn.Module):
    """module_utils.py OriginFloatLinear: plain float forward, keeps buf_* (used by deploy('origin_float'))."""

    def __init__(self, weight, bias, ori_module):
        super().__init__()
        self.register_buffer('weight', weight)
        if bias is not None:
            self.register_buffer('bias', bias)
        else:
            self.bias = None
        _copy_buf(self, ori_module)

    @torch.no_grad()
    def forward(self, x):
        return hip_linear(x, self.weight, self.bias)

    @classmethod
    @torch.no_grad()
    def new(cls, module):
        bias = module.bias.data if getattr(module, 'bias', None) is not None else None
        m = cls(module.weight.data, bias, module)
        m.in_features, m.out_features = module.in_features, module.out_features
        return m


class FakeQuantLinear(nn.Module):
    """module_utils.py:586-678: quantizes the weight lazily on first forward with w_qdq, activations with a_qdq."""

    def __init__(self, weight, bias, ori_module, w_qdq, a_qdq):
        super().__init__()
        self.register_buffer('weight', weight)
        if bias is not None:
            self.register_buffer('bias', bias)
        else:
            self.bias = None
        self.a_qdq = a_qdq
        self.w_qdq = w_qdq
        _copy_buf(self, ori_module)
        self.buf_rotate = False
        self.dynamic_quant_weight = False
        self.dynamic_quant_tmp_weight = False

    def forward(self, x):
        if self.a_qdq is not None:
            x = self.a_qdq(x, self)
        if not hasattr(self, 'tmp_weight'):
            self.register_buffer('tmp_weight', self.w_qdq(self), persistent=False)
            self.tmp_bias = self.bias
        elif self.dynamic_quant_weight or self.dynamic_quant_tmp_weight:
            self.tmp_weight = self.w_qdq(self)
            self.tmp_bias = self.bias
        return hip_linear(x, self.tmp_weight, self.tmp_bias)

    @classmethod
    @torch.no_grad()
    def new(cls, module, w_qdq, a_qdq):
        bias = module.bias.data if getattr(module, 'bias', None) is not None else None
        m = cls(module.weight.data, bias, ori_module=module, w_qdq=w_qdq, a_qdq=a_qdq)
        m.in_features, m.out_features = module.in_features, module.out_features
        m.w_qdq_name = _func_name(w_qdq)
        m.a_qdq_name = _func_name(a_qdq) if a_qdq is not None else 'None'
        return m

    @classmethod
    def get_func_name(cls, any_callable):
        return _func_name(any_callable)

    def __repr__(self):
        return (f'FakeQuantLinear(in_features={self.in_features},out_features={self.out_features}, '
                f'bias={self.bias is not None},weight_quant={self.w_qdq_name},act_quant={self.a_qdq_name})')


class EffcientFakeQuantLinear(nn.Module):
    """module_utils.py:681-759 (spelling as in the reference): the weight is fake-quantized once at new()."""

    def __init__(self, weight, bias, ori_module, a_qdq):
        super().__init__()
        self.register_buffer('weight', weight)
        if bias is not None:
            self.register_buffer('bias', bias)
        else:
            self.bias = None
        self.a_qdq = a_qdq
        _copy_buf(self, ori_module)
        self.buf_rotate = False

    @torch.no_grad()
    def forward(self, x):
        if self.a_qdq is not None:
            x = self.a_qdq(x, self)
        return hip_linear(x, self.weight, self.bias)

    @classmethod
    @torch.no_grad()
    def new(cls, module, w_qdq, a_qdq, debug_print={}):
        weight = w_qdq(module)
        bias = module.bias.data if module.bias is not None else None
        m = cls(weight, bias, ori_module=module, a_qdq=a_qdq)
        m.in_features, m.out_features = module.in_features, module.out_features
        m.w_qdq_name = _func_name(w_qdq)
        m.a_qdq_name = _func_name(a_qdq) if a_qdq is not None else 'None'
        m.debug_print = debug_print
        return m

    @classmethod
    def get_func_name(cls, any_callable):
        return _func_name(any_callable)

    def __repr__(self):
        return (f'EffcientFakeQuantLinear(in_features={self.in_features},out_features={self.out_features},'
                f'bias={self.bias is not None},weight_quant={self.w_qdq_name},act_quant={self.a_qdq_name})')


class VllmRealQuantLinear(nn.Module):
    """module_utils.py:762-876: compressed-tensors layout (weight_packed int32 [R, K*b/32] or weight int8/fp8,
    weight_scale, input_scale)."""

    def __init__(self, weight, bias, scales, input_scale, need_pack, scales_name):
        super().__init__()
        self.register_buffer('weight_packed' if need_pack else 'weight', weight)
        if bias is not None:
            self.register_buffer('bias', bias)
        else:
            self.bias = None
        self.register_buffer(scales_name, scales)
        self.register_buffer('input_scale', input_scale)

    @torch.no_grad()
    def forward(self, x):
        raise NotImplementedError

    @classmethod
    @torch.no_grad()
    def new(cls, module, w_q, quant_config):
        weight, scales = cls.quant_pack(module, w_q, quant_config)
        input_scale = getattr(module, 'buf_act_scales_0', None)
        if ('act' in quant_config and quant_config['act'].get('static', False)
                and quant_config.get('quant_type', 'int-quant') == 'int-quant'):
            input_scale = input_scale.unsqueeze(0)
        bias = module.bias.data if module.bias is not None else None
        need_pack = quant_config['weight'].get('need_pack', False)
        scales_name = 'weight_scale_inv' if quant_config['weight']['granularity'] == 'per_block' else 'weight_scale'
        m = cls(weight, bias, scales, input_scale, need_pack, scales_name)
        m.in_features, m.out_features = module.in_features, module.out_features
        m.weight_shape, m.weight_dtype = weight.shape, weight.dtype
        m.scales_shape, m.scales_dtype = scales.shape, scales.dtype
        m.zeros_shape = m.zeros_dtype = None
        return m

    @classmethod
    @torch.no_grad()
    def quant_pack(cls, module, w_q, quant_config):
        weight, scales, zeros = w_q(module)
        if quant_config['weight'].get('need_pack', False):
            weight, scales = cls.pack(weight, scales, quant_config)
        return weight, scales

    @classmethod
    @torch.no_grad()
    def pack(cls, weight, scales, quant_config):
        """module_utils.py:836-862 on the GPU: LSB-first nibbles/bytes of (code + 2^(b-1)) into int32."""
        return pack_lsb(weight, quant_config['weight']['bit']), scales.to(torch.float16)

    def __repr__(self):
        return (f'{type(self).__name__}(in_features={self.in_features}, out_features={self.out_features}, '
                f'bias={self.bias is not None}, weight_shape={self.weight_shape}, weight_dtype={self.weight_dtype}, '
                f'scales_shape={self.scales_shape}, scales_dtype={self.scales_dtype})')


class LightllmRealQuantLinear(VllmRealQuantLinear):
    pass


class Lightx2vRealQuantLinear(VllmRealQuantLinear):
    pass


class SglRealQuantLinear(VllmRealQuantLinear):
    pass


class AutoawqRealQuantLinear(nn.Module):
    """module_utils.py:936-1065: AutoAWQ GEMM layout (qweight [K, R/8], qzeros [K/g, R/8], scales [K/g, R] f16)."""

    def __init__(self, weight, bias, scales, zeros):
        super().__init__()
        self.register_buffer('qweight', weight)
        if bias is not None:
            self.register_buffer('bias', bias)
        else:
            self.bias = None
        self.register_buffer('scales', scales)
        if zeros is not None:
            self.register_buffer('qzeros', zeros)
        else:
            self.qzeros = None

    @torch.no_grad()
    def forward(self, x):
        raise NotImplementedError

    @classmethod
    @torch.no_grad()
    def new(cls, module, w_q, quant_config):
        weight, scales, zeros = cls.quant_pack(module, w_q, quant_config)
        bias = module.bias.data if module.bias is not None else None
        m = cls(weight, bias, scales, zeros)
        m.in_features, m.out_features = module.in_features, module.out_features
        m.weight_shape, m.weight_dtype = weight.shape, weight.dtype
        m.scales_shape, m.scales_dtype = scales.shape, scales.dtype
        m.zeros_shape = zeros.shape if zeros is not None else None
        m.zeros_dtype = zeros.dtype if zeros is not None else None
        return m

    @classmethod
    @torch.no_grad()
    def quant_pack(cls, module, w_q, quant_config):
        _, scales, zeros = w_q(module)
        pack_version = quant_config['weight']['pack_version']
        if pack_version != 'gemm_pack':
            raise NotImplementedError(f'Not support {pack_version}.')
        return cls.gemm_pack(module, module.weight.data, scales, zeros, quant_config)

    @classmethod
    @torch.no_grad()
    def gemm_pack(cls, module, weight, scales, zeros, quant_config):
        assert scales is not None and zeros is not None
        if quant_config['weight']['bit'] != 4:
            raise NotImplementedError('Only 4-bit are supported for now.')
        return pack_awq_gemm(weight, scales, zeros, quant_config['weight']['group_size'])


class MlcllmRealQuantLinear(AutoawqRealQuantLinear):
    pass


class LlmcFp8Linear(nn.Module):
    """module_utils.py:130-191: a Linear whose checkpoint weight is block-scaled FP8 (DeepSeek-V3): `weight`
    float8_e4m3fn [R, K] + `weight_scale_inv` fp32 [ceil(R/b), ceil(K/b)]. forward quantizes the activation per
    1 x b block and runs the block-scaled fp8 GEMM (kernel.block_wise_fp8_forward_func) — on gfx950 the HIP kernels
    replace the reference's Hopper-only Triton path, so its bf16 de-quantize-and-F.linear fallback is not needed."""

    def __init__(self, in_features, out_features, bias, block_size):
        super().__init__()
        self.block_size, self.in_features, self.out_features = block_size, in_features, out_features
        if bias is not None:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=torch.float8_e4m3fn), requires_grad=False)
        so, si = -(-out_features // block_size), -(-in_features // block_size)
        self.weight_scale_inv = nn.Parameter(torch.empty(so, si, dtype=torch.float32), requires_grad=False)

    def forward(self, x):
        from .kernel import block_wise_fp8_forward_func, weight_cast_to_bf16
        if self.weight.data.dtype == torch.float8_e4m3fn:
            if self.block_size == 128 and x.shape[-1] % 128 == 0:
                return block_wise_fp8_forward_func(x, self.weight.data, self.weight_scale_inv.data, self.block_size,
                                                   None if self.bias is None else self.bias.data)
            w = weight_cast_to_bf16(self.weight.data, self.weight_scale_inv.data, self.block_size)
            return hip_linear(x, w, self.bias)
        return hip_linear(x, self.weight, self.bias)

    @classmethod
    @torch.no_grad()
    def new(cls, module, block_size):
        return cls(module.in_features, module.out_features, module.bias, block_size)

    def __repr__(self):
        return (f'LlmcFp8Linear(in_features={self.in_features}, out_features={self.out_features}, '
                f'bias={self.bias is not None}, weight_shape={self.weight.shape}, weight_dtype={self.weight.dtype}, '
                f'block_size={self.block_size})')


_TRANSFORMERS_LINEAR_TYPES_ = [nn.Linear]
_TRANSFORMERS_LN_TYPES_ = [nn.LayerNorm]
try:  # RMSNorm-style layers of HF models count as LN types for apply_scale
    from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS
    _TRANSFORMERS_LN_TYPES_ = list(ALL_LAYERNORM_LAYERS)
except Exception:  # pragma: no cover
    pass
if hasattr(nn, 'RMSNorm') and nn.RMSNorm not in _TRANSFORMERS_LN_TYPES_:
    _TRANSFORMERS_LN_TYPES_.append(nn.RMSNorm)

_LLMC_LN_TYPES_ = []
_LLMC_LINEAR_TYPES_ = [LlmcFp8Linear, OriginFloatLinear, FakeQuantLinear, EffcientFakeQuantLinear, VllmRealQuantLinear,
                       SglRealQuantLinear, AutoawqRealQuantLinear, MlcllmRealQuantLinear, LightllmRealQuantLinear]
_REALQUANT_LINEAR_MAP_ = {
    'vllm_quant': VllmRealQuantLinear,
    'lightllm_quant': LightllmRealQuantLinear,
    'sgl_quant': SglRealQuantLinear,
    'autoawq_quant': AutoawqRealQuantLinear,
    'mlcllm_quant': MlcllmRealQuantLinear,
    'lightx2v_quant': Lightx2vRealQuantLinear,
}
