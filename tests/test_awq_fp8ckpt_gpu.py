"""AWQ on block-wise FP8 checkpoints (DeepSeek-V3 layout; awq.py:53-58, 147-164, base_blockwise_quantization.py:46-68,
655-700, 750-775) on MI355X: our classes on LlmcFp8Linear modules against the reference's own class code
(tests/golden/awq_fp8ckpt.npz, non-Triton casts) and, for the kernel (Triton-arithmetic) casts, against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import awq_ref as A
from oracle import quant_ref as Q

pytestmark = pytest.mark.gpu


def _awq(bit, sym, gs, bsz, cast):
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.awq import Awq
    a = Awq.__new__(Awq)
    a.wquantizer = (IntegerQuantizer(bit, bool(sym), 'per_group', group_size=gs) if gs
                    else IntegerQuantizer(bit, bool(sym), 'per_channel'))
    a.fp8_block_size, a.fp8_cast, a.fp8_cast_semantics = bsz, cast, 'qtorch'
    a.has_gqa = a.do_gqa_trans = False
    return a


def _layer(b8, s8, bsz):
    from llmc_amd.compression.quantization.module_utils import LlmcFp8Linear
    R, K = b8.shape
    l = LlmcFp8Linear(K, R, None, bsz).cuda()
    l.weight.data = torch.from_numpy(np.ascontiguousarray(b8)).cuda().view(torch.float8_e4m3fn)
    l.weight_scale_inv.data = torch.from_numpy(np.ascontiguousarray(s8)).cuda()
    return l


def _same(layer, b8, s8, msg):
    assert layer.weight.data.dtype == torch.float8_e4m3fn and layer.weight_scale_inv.data.dtype == torch.float32, msg
    np.testing.assert_array_equal(layer.weight_scale_inv.data.cpu().numpy().view(np.uint32), np.ascontiguousarray(s8).view(np.uint32), err_msg=msg)
    np.testing.assert_array_equal(layer.weight.data.view(torch.uint8).cpu().numpy(), b8, err_msg=msg)


def test_fp8_checkpoint_branches_bit_exact_vs_reference_golden():
    """special.fp8_cast = quantizer (the reference's non-Triton binding, awq.py:17-20): get_weight_scale,
    fake_quantize_weight, w_qdq, scale_ln_fcs and scale_fc_fc leave exactly the reference's codes and block scales."""
    g = load_golden('awq_fp8ckpt')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        bit, sym, gs, bsz, K = [int(v) for v in g[p + 'meta']]
        a = _awq(bit, sym, gs, bsz, 'quantizer')
        layers = [_layer(g[p + f'w8_{i}'], g[p + f's8_{i}'], bsz) for i in range(2)]
        w_max = a.get_weight_scale({f'l{i}': l for i, l in enumerate(layers)})
        assert w_max.dtype == torch.bfloat16
        np.testing.assert_array_equal(w_max.float().cpu().numpy().view(np.uint32), g[p + 'w_max'].view(np.uint32), err_msg=name)
        cols = torch.from_numpy(g[p + 'scales']).to(torch.bfloat16).cuda()
        for i, l in enumerate(layers):
            w0, s0 = l.weight.data.clone(), l.weight_scale_inv.data.clone()
            w8, s8 = a._fake_quantize_weight(w0, cols, s0)
            assert torch.equal(w0.view(torch.uint8), l.weight.data.view(torch.uint8))       # the original is not modified
            np.testing.assert_array_equal(s8.cpu().numpy().view(np.uint32), g[p + f'fq_s8_{i}'].view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(w8.view(torch.uint8).cpu().numpy(), g[p + f'fq_w8_{i}'], err_msg=name)
            r = a.w_qdq(l, a.wquantizer)
            np.testing.assert_array_equal(r.view(torch.uint8).cpu().numpy(), g[p + f'qdq_w8_{i}'], err_msg=name)
            np.testing.assert_array_equal(l.weight_scale_inv.data.cpu().numpy().view(np.uint32), g[p + f'qdq_s8_{i}'].view(np.uint32), err_msg=name)
            l.weight.data, l.weight_scale_inv.data = w0, s0
        ln = torch.nn.LayerNorm(K).to(torch.bfloat16).cuda()
        ln.weight.data = torch.from_numpy(g[p + 'ln_w']).to(torch.bfloat16).cuda()
        ln.bias.data = torch.from_numpy(g[p + 'ln_b']).to(torch.bfloat16).cuda()
        a.scale_ln_fcs(ln, layers, cols)
        np.testing.assert_array_equal(ln.weight.data.float().cpu().numpy().view(np.uint32), g[p + 'ln_w_after'].view(np.uint32))
        np.testing.assert_array_equal(ln.bias.data.float().cpu().numpy().view(np.uint32), g[p + 'ln_b_after'].view(np.uint32))
        for i, l in enumerate(layers):
            _same(l, g[p + f'ln_w8_{i}'], g[p + f'ln_s8_{i}'], name)
        fc1, fc2 = _layer(g[p + 'fc1_w8'], g[p + 'fc1_s8'], bsz), _layer(g[p + 'fc2_w8'], g[p + 'fc2_s8'], bsz)
        a.scale_fc_fc(fc1, fc2, cols)
        _same(fc1, g[p + 'fc1_w8_after'], g[p + 'fc1_s8_after'], name)
        _same(fc2, g[p + 'fc2_w8_after'], g[p + 'fc2_s8_after'], name)


def test_fp8_checkpoint_branches_with_the_kernel_casts_match_the_oracle():
    """The default binding (special.fp8_cast = kernel: the arithmetic of the reference's Triton casts, which it binds on
    FP8-capable GPUs, awq.py:14-16): same chain with the e4m3fn cast instead of qtorch's rounding."""
    g = load_golden('awq_fp8ckpt')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        bit, sym, gs, bsz, K = [int(v) for v in g[p + 'meta']]
        if bsz != 128:
            continue                     # the reference's Triton casts are 128 x 128 kernels
        qmin, qmax = Q.int_range(bit, bool(sym))
        a = _awq(bit, sym, gs, bsz, 'kernel')
        cols_np = g[p + 'scales']
        cols = torch.from_numpy(cols_np).to(torch.bfloat16).cuda()
        for i in range(2):
            b8, s8 = g[p + f'w8_{i}'], g[p + f's8_{i}']
            l = _layer(b8, s8, bsz)
            w8, sc = a._fake_quantize_weight(l.weight.data, cols, l.weight_scale_inv.data)
            rb, rs = A.fp8ckpt_fake_quantize_weight(b8, s8, cols_np, bsz, bool(sym), qmin, qmax, gs, sem='cast')
            np.testing.assert_array_equal(sc.cpu().numpy().view(np.uint32), rs.view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(w8.view(torch.uint8).cpu().numpy(), rb, err_msg=name)


def test_search_on_fp8_checkpoint_layers_runs_the_general_route_and_restores_the_modules():
    """Awq.search_scale_subset over two LlmcFp8Linear layers: every grid point evaluates the re-blocked FP8 weights
    through the layers' own block-scaled forward; the winner equals a brute-force replay of awq.py:189-253 with the same
    primitives, and weights AND block scales are back afterwards."""
    from llmc_amd.compression.quantization import awq_ops
    from llmc_amd.compression.quantization.awq import Awq
    g = load_golden('awq_fp8ckpt')
    p = 'w4g128_asym/'
    bit, sym, gs, bsz, K = [int(v) for v in g[p + 'meta']]
    a = _awq(bit, sym, gs, bsz, 'kernel')
    a.w_only, a.awq_bs, a.save_mem, a.padding_mask, a.trans_version, a.n_samples = True, None, False, None, 'v2', 2
    layers = [_layer(g[p + f'w8_{i}'], g[p + f's8_{i}'], bsz) for i in range(2)]

    class Stacked(torch.nn.Module):
        def __init__(self, ls):
            super().__init__()
            self.ls = torch.nn.ModuleList(ls)

        def forward(self, x):
            return torch.cat([l(x) for l in self.ls], dim=-1)

    mod = Stacked(layers)
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 96, K, generator=gen) * torch.exp(0.8 * torch.randn(K, generator=gen))).to(torch.bfloat16).cuda()
    keep = [(l.weight.data.clone(), l.weight_scale_inv.data.clone()) for l in layers]
    ld = {f'l{i}': l for i, l in enumerate(layers)}
    best = Awq.search_scale_subset(a, None, ld, [x], mod, False, {})
    for l, (w0, s0) in zip(layers, keep):
        assert torch.equal(l.weight.data.view(torch.uint8), w0.view(torch.uint8)) and torch.equal(l.weight_scale_inv.data, s0)
    # replay
    org = mod(x)
    xm = a._act_scale_batched(x)
    w_max = a.get_weight_scale(ld)
    losses, cands = [], []
    for n in range(20):
        s = awq_ops.awq_scales(xm, w_max, n / 20, 'v2')
        for l, (w0, s0) in zip(layers, keep):
            l.weight.data, l.weight_scale_inv.data = a._fake_quantize_weight(w0, s, s0)
        out = mod(awq_ops.div_cols(x, s))
        losses.append(float((org - out).float().pow(2).mean()))
        cands.append(s)
        for l, (w0, s0) in zip(layers, keep):
            l.weight.data, l.weight_scale_inv.data = w0, s0
    assert np.isfinite(losses).all() and len(set(losses)) > 10
    assert torch.equal(best, cands[int(np.argmin(losses))])


def test_auto_clipper_runs_on_fp8_checkpoint_layers_deblocked_and_reblocks_them():
    """AutoClipper.run on block-wise FP8 weights (auto_clip.py:47-53, 78-81): the weight is de-blocked to bf16 for
    the clip search and the clamp and re-blocked afterwards — equal to doing the three steps by hand on a bf16 Linear; q / k
    layers are skipped and keep their FP8 weights. (The reference itself raises AttributeError here: its AutoClipper never
    receives fp8_block_size, so there is no golden; the three steps are each pinned on their own.)"""
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.auto_clip import AutoClipper
    g = load_golden('awq_fp8ckpt')
    p = 'w4g128_asym/'
    bit, sym, gs, bsz, K = [int(v) for v in g[p + 'meta']]
    a = _awq(bit, sym, gs, bsz, 'kernel')
    wq = IntegerQuantizer(4, False, 'per_group', group_size=128)
    clipper = AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v1', clip_sym=False, save_clip=False,
                          padding_mask=None)
    clipper.fp8_block_size, clipper.fp8_to_bf16, clipper.bf16_to_fp8 = bsz, a._fp8_to_bf16, a._bf16_to_fp8
    block = torch.nn.Module()
    block.o_proj = _layer(g[p + 'w8_0'], g[p + 's8_0'], bsz)
    block.k_proj = _layer(g[p + 'w8_1'], g[p + 's8_1'], bsz)
    keep_k = (block.k_proj.weight.data.clone(), block.k_proj.weight_scale_inv.data.clone())
    gen = torch.Generator().manual_seed(9)
    x = (torch.randn(2, 96, K, generator=gen) * torch.exp(0.8 * torch.randn(K, generator=gen))).to(torch.bfloat16).cuda()
    feat = {'o_proj': [x], 'k_proj': [x]}
    # by hand
    w_bf16 = a._fp8_to_bf16(block.o_proj.weight, block.o_proj.weight_scale_inv)
    assert w_bf16.dtype == torch.bfloat16
    plain = torch.nn.Linear(K, w_bf16.shape[0], bias=False).cuda()
    plain.weight.data = w_bf16.clone()
    mx, mn = clipper.auto_clip_layer(0, 'o_proj', plain.weight, [x], n_sample_token=96)
    clipper.apply_clip(0, plain, mn, mx, 'o_proj')
    assert not torch.equal(plain.weight.data, w_bf16)                      # something was clipped
    w8, s8 = a._bf16_to_fp8(plain.weight.data)
    clipper.run(block, 0, feat, n_sample_token=96)
    assert block.o_proj.weight.data.dtype == torch.float8_e4m3fn and block.o_proj.weight_scale_inv.data.dtype == torch.float32
    assert torch.equal(block.o_proj.weight.data.view(torch.uint8), w8.view(torch.uint8))
    assert torch.equal(block.o_proj.weight_scale_inv.data, s8)
    assert torch.equal(block.k_proj.weight.data.view(torch.uint8), keep_k[0].view(torch.uint8))
    assert torch.equal(block.k_proj.weight_scale_inv.data, keep_k[1])
    y = block.o_proj(x)                                                    # the re-blocked layer still runs its fp8 forward
    assert y.shape == (2, 96, w_bf16.shape[0]) and bool(torch.isfinite(y.float()).all())
