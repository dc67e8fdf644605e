"""§8(f) rank 4 — FP8 block-wise (DeepSeek-V3 layout) on MI355X: FloatQuantizer `per_block` and the reference's
weight_cast_to_fp8 / weight_cast_to_bf16 bit-exact against goldens produced by the reference's own (non-Triton) code;
act_quant / weight_cast_to_fp8 / fp8_gemm against goldens produced by the reference's own Triton kernels run unmodified
on an MI355X (tests/golden/fp8_triton.npz, tools/fp8_triton_golden.py) and against the restatement of kernel.py in
oracle/quant_ref.py, which those goldens pin (tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import quant_ref as Q

pytestmark = pytest.mark.gpu
TD = {'f16': torch.float16, 'bf16': torch.bfloat16}


@pytest.mark.parametrize('golden,sem', [('fp8_block_qtorch', 'qtorch'), ('fp8_block', 'cast')])
def test_per_block_quantizer_and_weight_casts_match_reference_golden(golden, sem):
    """fp8_block_qtorch.npz: the reference's class with float_quantize = the restated qtorch (FloatQuantizer's default here);
    fp8_block.npz: bound to torch's e4m3fn cast (fp8_semantics='cast', the Triton kernels' arithmetic)."""
    from llmc_amd.compression.quantization import FloatQuantizer
    from llmc_amd.compression.quantization.quant import weight_cast_to_bf16, weight_cast_to_fp8
    g = load_golden(golden)
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        dt, b = TD[str(g[p + 'dt'])], int(g[p + 'block'])
        w = torch.from_numpy(g[p + 'w']).to(dt).cuda()
        q = FloatQuantizer('e4m3', True, 'per_block', block_size=b, use_qtorch=True, fp8_semantics=sem)
        rw, rs, rz = q.real_quant_weight_dynamic(w)
        assert rw.dtype == torch.float8_e4m3fn and rz is None and rs.dtype == torch.float32
        assert tuple(rs.shape) == g[p + 'scales'].shape
        np.testing.assert_array_equal(rs.cpu().numpy().view(np.uint32), g[p + 'scales'].view(np.uint32), err_msg=str(ci))
        np.testing.assert_array_equal(rw.view(torch.uint8).cpu().numpy(), g[p + 'bits'], err_msg=str(ci))
        fk = q.fake_quant_weight_dynamic(w)
        assert fk.dtype == dt
        np.testing.assert_array_equal(fk.float().cpu().numpy().view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(ci))
        t, s4, z, qmax, qmin = q.get_tensor_qparams(w)
        assert t.dim() == 4 and s4.shape == (rs.shape[0], 1, rs.shape[1], 1) and float(qmax) == 448.0
        # static forms with the scales just found reproduce the dynamic results
        assert torch.equal(q.fake_quant_weight_static(w, {'scales': s4}), fk)
        assert torch.equal(q.real_quant_weight_static(w, {'scales': s4})[0].view(torch.uint8), rw.view(torch.uint8))
        w8, s8 = weight_cast_to_fp8(w, b, fp8_semantics=sem)
        np.testing.assert_array_equal(w8.view(torch.uint8).cpu().numpy(), g[p + 'cast_bits'])
        np.testing.assert_array_equal(s8.cpu().numpy().view(np.uint32), g[p + 'cast_scales'].view(np.uint32))
        back = weight_cast_to_bf16(w8, s8, b)
        assert back.dtype == torch.bfloat16
        np.testing.assert_array_equal(back.float().cpu().numpy().view(np.uint32), g[p + 'cast_back'].view(np.uint32))


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
def test_act_quant_vs_restatement(dt):
    from llmc_amd.compression.quantization.kernel import act_quant
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(5, 37, 512, generator=gen) * torch.exp(torch.randn(512, generator=gen))).to(TD[dt])
    x[0, 0, :128] = 0.0                       # an all-zero block: scale 0, 0 / 0 = NaN codes like the Triton kernel
    y, s = act_quant(x.cuda(), 128)
    assert y.dtype == torch.float8_e4m3fn and s.shape == (5, 37, 4) and s.dtype == torch.float32
    bits, sref = Q.act_quant_ref(x.float().numpy(), 128)
    np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), sref.view(np.uint32))
    np.testing.assert_array_equal(y.view(torch.uint8).cpu().numpy(), bits)
    assert float(s[0, 0, 0]) == 0.0 and (y.view(torch.uint8)[0, 0, :128] & 0x7f == 0x7f).all()


@pytest.mark.parametrize('shape', [(256, 256, 512), (300, 200, 384), (64, 1024, 1024), (130, 136, 200)])
def test_fp8_block_gemm_and_forward_vs_restatement(shape):
    from llmc_amd.compression.quantization import kernel as KN
    M, N, K = shape
    gen = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=gen) * torch.exp(0.5 * torch.randn(K, generator=gen))).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=gen).to(torch.bfloat16).cuda()
    if K % 128:
        # act_quant needs K % block == 0 (kernel.py:46-48); exercise the GEMM's ragged K with hand-made operands
        a8 = (torch.randn(M, K, generator=gen) * 3).to(torch.float8_e4m3fn).cuda()
        nkb = -(-K // 128)
        a_s = torch.rand(M, nkb, generator=gen).cuda() + 0.5
    else:
        a8, a_s = KN.act_quant(x, 128)
    w8, w_s = KN.weight_cast_to_fp8(w, 128)
    c = KN.fp8_gemm(a8, a_s, w8, w_s)
    assert c.dtype == torch.bfloat16 and c.shape == (M, N)
    ref = Q.fp8_block_gemm_ref(a8.view(torch.uint8).cpu().numpy(), a_s.cpu().numpy(), w8.view(torch.uint8).cpu().numpy(),
                               w_s.cpu().numpy())
    refb = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
    err = np.abs(c.float().cpu().numpy() - refb)
    tol = 2.0 ** -7 * np.abs(ref) + 1e-4 * np.abs(ref).max()        # one bf16 rounding of sums that differ in fp32 order
    assert (err <= tol).all(), float((err - tol).max())
    if K % 128 == 0:
        y = KN.block_wise_fp8_forward_func(x, w8, w_s, 128, bias)
        want = (torch.from_numpy(ref).to(torch.bfloat16).cuda() + bias)
        d = (y.float() - want.float()).abs()
        assert bool((d <= 2.0 ** -6 * want.float().abs() + 2e-4 * float(np.abs(ref).max()) + 2.0 ** -7 * bias.float().abs()).all())
        # sanity against the unquantized product: FP8 block-wise is a ~3 % approximation of x W^T
        full = x.float() @ w.float().T + bias.float()
        rel = ((y.float() - full).norm() / full.norm()).item()
        assert rel < 0.06, rel


@pytest.mark.parametrize('shape,dtype,with_bias', [
    ((512, 768, 1024), torch.bfloat16, False),      # whole 256 x 256 tiles, even number of K blocks
    ((384, 256, 640), torch.float16, False),        # ragged m tile, odd number of K blocks
    ((1000, 520, 384), torch.float32, True),        # ragged both ways, three K blocks, bias, fp32 output
    ((128, 130, 128), torch.bfloat16, True),        # the smallest shape the K = 64 kernel takes, N % 4 != 0
    ((2048, 2304, 256), torch.bfloat16, False),     # 72 tiles: several rounds of the persistent grid's XCD order
])
def test_fp8_block_gemm_k64_kernel(shape, dtype, with_bias):
    """The 256 x 256-tile kernel on v_mfma_f32_32x32x64_f8f6f4 (K % 128 == 0, M, N >= 128) against the fp64 restatement of
    kernel.py:213-242, for every output dtype, edge tiles, bias; operands with an asymmetric pattern so that a swapped or
    transposed tile cannot pass."""
    from llmc_amd.compression.quantization import kernel as KN
    M, N, K = shape
    gen = torch.Generator().manual_seed(M * 7 + N)
    a8 = (torch.randn(M, K, generator=gen) * 2 * (1 + torch.arange(M).remainder(7)[:, None] / 7)).to(torch.float8_e4m3fn).cuda()
    w8 = (torch.randn(N, K, generator=gen) * (1 + torch.arange(N).remainder(5)[:, None] / 5)).to(torch.float8_e4m3fn).cuda()
    nkb = K // 128
    a_s = (torch.rand(M, nkb, generator=gen) + 0.5).cuda()
    w_s = (torch.rand(-(-N // 128), nkb, generator=gen) * 0.1 + 0.01).cuda()
    bias = torch.randn(N, generator=gen).to(dtype).cuda() if with_bias else None
    c = KN.fp8_gemm(a8, a_s, w8, w_s, dtype=dtype, bias=bias)
    assert c.dtype == dtype and c.shape == (M, N)
    ref = Q.fp8_block_gemm_ref(a8.view(torch.uint8).cpu().numpy(), a_s.cpu().numpy(), w8.view(torch.uint8).cpu().numpy(),
                               w_s.cpu().numpy())
    eps = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 2.0 ** -20}[dtype]
    want = torch.from_numpy(ref).to(dtype)
    if with_bias:
        want = (want + bias.cpu()).to(dtype)
    err = (c.cpu().double() - want.double()).abs().numpy()
    # one output rounding of fp32 sums taken in another order (in-block: the MFMA's; fp32 output: 128 terms of rounding noise)
    tol = eps * np.abs(want.double().numpy()) + (3e-5 if dtype == torch.float32 else 1e-4) * np.abs(ref).max()
    if with_bias:
        tol = tol + eps * np.abs(ref)          # the product is rounded to the output dtype before the bias is added
    assert (err <= tol).all(), (float((err - tol).max()), np.unravel_index(np.argmax(err - tol), err.shape))
    # the opt-in one-fma update (fused_scale: the scale product rounded once per row and K block) stays inside the same envelope and
    # within a few fp32 roundings per K block of the bit-identical form
    f = KN.fp8_gemm(a8, a_s, w8, w_s, dtype=dtype, bias=bias, fused_scale=True)
    errf = (f.cpu().double() - want.double()).abs().numpy()
    assert (errf <= tol).all(), float((errf - tol).max())
    if dtype == torch.float32 and not with_bias:
        assert float((f - c).abs().max()) <= 4e-7 * (K // 128) * float(c.abs().max())


def test_llmc_fp8_linear_forward():
    """LlmcFp8Linear (module_utils.py:130-191) loaded with a block-scaled FP8 weight: forward = act_quant + fp8 GEMM."""
    from llmc_amd.compression.quantization import LlmcFp8Linear
    from llmc_amd.compression.quantization import kernel as KN
    gen = torch.Generator().manual_seed(9)
    lin = torch.nn.Linear(512, 384, bias=True).to(torch.bfloat16)
    m = LlmcFp8Linear.new(lin, 128).cuda()
    w = (torch.randn(384, 512, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    w8, ws = KN.weight_cast_to_fp8(w, 128)
    m.weight.data, m.weight_scale_inv.data = w8, ws
    m.bias.data = torch.randn(384, generator=gen).to(torch.bfloat16).cuda()
    x = torch.randn(4, 33, 512, generator=gen).to(torch.bfloat16).cuda()
    y = m(x)
    assert y.shape == (4, 33, 384) and y.dtype == torch.bfloat16
    full = x.float() @ w.float().T + m.bias.data.float()
    assert ((y.float() - full).norm() / full.norm()).item() < 0.06
    assert 'LlmcFp8Linear' in repr(m) and m.weight.dtype == torch.float8_e4m3fn


def _from16(bits, dt):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(TD[dt])


def test_triton_only_ops_match_goldens_of_the_reference_triton_kernels():
    """kernel.py:7-242 as executed by the reference itself (Triton on ROCm, this GPU model): our HIP act_quant and
    weight_cast_to_fp8 give the same fp8 codes and fp32 scales bit for bit (the 0 / 0 block included); so does the block-scaled
    GEMM's bf16 output."""
    from llmc_amd.compression.quantization import kernel as KN
    g = load_golden('fp8_triton')
    for dt in ('bf16', 'f16'):
        x = _from16(g[f'aq_{dt}_x16'], dt).cuda()
        y, s = KN.act_quant(x, 128)
        np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), g[f'aq_{dt}_scales'].view(np.uint32))
        np.testing.assert_array_equal(y.view(torch.uint8).cpu().numpy(), g[f'aq_{dt}_bits'])
    for i in range(int(g['n_gemm'])):
        p = f'g{i}_'
        x, w = _from16(g[p + 'x16'], 'bf16').cuda(), _from16(g[p + 'w16'], 'bf16').cuda()
        a8, a_s = KN.act_quant(x, 128)
        w8, w_s = KN.weight_cast_to_fp8(w, 128)
        np.testing.assert_array_equal(a8.view(torch.uint8).cpu().numpy(), g[p + 'a_bits'])
        np.testing.assert_array_equal(a_s.cpu().numpy().view(np.uint32), g[p + 'a_s'].view(np.uint32))
        np.testing.assert_array_equal(w8.view(torch.uint8).cpu().numpy(), g[p + 'w_bits'])
        np.testing.assert_array_equal(w_s.cpu().numpy().view(np.uint32), g[p + 'w_s'].view(np.uint32))
        c = KN.fp8_gemm(a8, a_s, w8, w_s).float().cpu().numpy()
        cref = _from16(g[p + 'c_bf16_16'], 'bf16').float().numpy()
        assert (np.abs(c - cref) <= 2.0 ** -7 * np.abs(cref) + 1e-4 * np.abs(cref).max()).all()
        from conftest import report
        report(f'fp8_triton_gemm/{i}', equal_fraction=float((c == cref).mean()))
        # the K = 64 MFMA and fma(part * a_s, b_s, acc) are what the Triton kernel compiles to on this GPU: every output equal
        # (measured 1.0 on both golden products; 0.97-0.99 with the K = 16 MFMA and an uncontracted update)
        assert (c == cref).all(), float((c == cref).mean())


def test_fp8_gemm_bit_identical_to_the_reference_triton_kernel_more_shapes():
    """fp8_triton_more.npz (tools/fp8_triton_golden_more.py: the reference's Triton fp8_gemm run unmodified on an MI355X):
    several b_s column blocks, up to 12 K blocks, M below one tile (the K = 16 MFMA kernel), ragged M / N — bf16 AND fp32
    outputs bit for bit (measured: 1.0 on every shape, also on six larger ones up to 1000 x 520 x 2048)."""
    from conftest import report
    from llmc_amd.compression.quantization import kernel as KN
    g = load_golden('fp8_triton_more')
    for i in range(int(g['n_gemm'])):
        p = f'g{i}_'
        M, N, Kd = [int(v) for v in g[p + 'shape']]
        a8 = torch.from_numpy(g[p + 'a_bits']).cuda().view(torch.float8_e4m3fn)
        w8 = torch.from_numpy(g[p + 'w_bits']).cuda().view(torch.float8_e4m3fn)
        a_s, w_s = torch.from_numpy(g[p + 'a_s']).cuda(), torch.from_numpy(g[p + 'w_s']).cuda()
        c = KN.fp8_gemm(a8, a_s, w8, w_s, dtype=torch.bfloat16)
        c32 = KN.fp8_gemm(a8, a_s, w8, w_s, dtype=torch.float32)
        ref32 = torch.from_numpy(g[p + 'c_f32'])
        eq16 = float((c.cpu() == ref32.to(torch.bfloat16)).float().mean())      # Triton's bf16 output = its fp32 one rounded (checked when the golden was made)
        eq32 = float((c32.cpu().view(torch.int32) == ref32.view(torch.int32)).float().mean())
        report(f'fp8_triton_more/{M}x{N}x{Kd}', bf16_equal=eq16, f32_equal=eq32)
        assert eq16 == 1.0 and eq32 == 1.0, (M, N, Kd, eq16, eq32)


def test_fp8_per_group_matches_reference_golden():
    """FloatQuantizer per_group (FP8 activations in groups of 128 / 32, e4m3 and e5m2; rtn_w_a_block.yml) against the reference's class."""
    import numpy as np
    import torch
    from conftest import load_golden
    from llmc_amd.compression.quantization import FloatQuantizer
    g = load_golden('fp8_group_qtorch')
    TDT = {'f16': torch.float16, 'bf16': torch.bfloat16}
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        dt, bit, gs = str(g[p + 'dt']), str(g[p + 'bit']), int(g[p + 'gs'])
        q = FloatQuantizer(bit, True, 'per_group', group_size=gs, use_qtorch=True)
        x = torch.from_numpy(g[p + 'x']).to(TDT[dt]).cuda()
        w = torch.from_numpy(g[p + 'w']).to(TDT[dt]).cuda()
        fa = q.fake_quant_act_dynamic(x).float().cpu().numpy()
        fw = q.fake_quant_weight_dynamic(w).float().cpu().numpy()
        np.testing.assert_array_equal(fa.view(np.uint32), g[p + 'fake_x'].view(np.uint32), err_msg=f'{ci} act')
        np.testing.assert_array_equal(fw.view(np.uint32), g[p + 'fake_w'].view(np.uint32), err_msg=f'{ci} weight')


def test_per_block_qtorch_subnormal_midpoints_follow_qpytorch_double_rounding():
    """qtorch rounds fl32(|x| + 2^-6) first and the sum's bits second; a quotient 1..16 fp32 ulps below a
    (k + 0.5) 2^-9 midpoint of e4m3's subnormal range is carried over the midpoint by the first rounding. FloatQuantizer per_block
    feeds fp32 quotients w / s to that routine: an fp32 weight whose block absmax is 448 (s = 1) and whose other elements sit
    around those midpoints must come out with the codes of the restated qtorch, element for element."""
    from llmc_amd.compression.quantization import FloatQuantizer
    ks = np.arange(0, 8, dtype=np.float32)
    mids = (ks + 0.5) * np.float32(2.0 ** -9)
    vals = []
    for m in mids:
        b = np.array([m], dtype=np.float32).view(np.uint32)[0]
        for d in range(-20, 21):
            vals.append(np.array([np.uint32(int(b) + d)], dtype=np.uint32).view(np.float32)[0])
    vals = np.array(vals, dtype=np.float32)
    w = np.zeros((128, 128), dtype=np.float32)
    w.ravel()[:vals.size] = vals
    w.ravel()[vals.size:2 * vals.size] = -vals
    w[127, 127] = 448.0                                   # absmax 448: the block scale is exactly 1
    q = FloatQuantizer('e4m3', True, 'per_block', block_size=128, use_qtorch=True)
    rw, rs, _ = q.real_quant_weight_dynamic(torch.from_numpy(w).cuda())
    assert float(rs.reshape(-1)[0]) == 1.0
    ref = Q.qtorch_float_quantize(w, 4, 3)
    got = rw.float().cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    # and the restatement really does cross those midpoints (the case the single-rounding form got wrong)
    x = np.array([np.float32(1.5 * 2.0 ** -9)], dtype=np.float32).view(np.uint32)
    below = np.array([x[0] - 1], dtype=np.uint32).view(np.float32)
    assert float(Q.qtorch_float_quantize(below, 4, 3)[0]) == 2.0 ** -8
