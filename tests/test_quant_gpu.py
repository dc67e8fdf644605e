"""HIP quantizer kernels (through the Python operator surface -> C ABI) vs the numpy oracle: bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import quant_ref as Q

pytestmark = pytest.mark.gpu

TD = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}


def dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(TD[dt]).cuda()


def host(t):
    return t.detach().float().cpu().numpy()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_quantizer(bit, sym, gran, gs):
    from llmc_amd.compression.quantization import IntegerQuantizer
    kw = dict(group_size=gs) if gs else {}
    return IntegerQuantizer(int(bit), bool(sym), gran, **kw)


def test_golden_cases_through_hip():
    g = load_golden('quant')
    for ci in range(int(g['n_cases'])):
        p = f'c{ci}_'
        bit, sym, gs, qmin, qmax = g[p + 'meta']
        dt, gran = str(g[p + 'dt']), str(g[p + 'gran'])
        q = make_quantizer(bit, sym, gran, int(gs))
        w = dev(g[p + 'w'], dt)
        tag = f'case {ci}: {dt} bit={bit} sym={sym} {gran} g={gs}'
        _, s, z, _, _ = q.get_tensor_qparams(w)
        assert s.dtype == TD[dt]
        np.testing.assert_array_equal(bits(host(s).reshape(-1)), bits(g[p + 'scales']), err_msg=tag)
        if not sym:
            np.testing.assert_array_equal(host(z).reshape(-1), g[p + 'zeros'], err_msg=tag)
        fq = q.fake_quant_weight_dynamic(w)
        assert fq.dtype == TD[dt] and fq.shape == w.shape
        np.testing.assert_array_equal(bits(host(fq)), bits(g[p + 'fake']), err_msg=tag)
        codes, rs, rz = q.real_quant_weight_dynamic(w)
        assert str(codes.dtype) == str(g[p + 'codes_dtype']), tag
        np.testing.assert_array_equal(codes.cpu().numpy().astype(np.int32), g[p + 'codes'], err_msg=tag)
        np.testing.assert_array_equal(bits(host(rs)), bits(g[p + 'rscales']), err_msg=tag)
        if not sym:
            np.testing.assert_array_equal(rz.cpu().numpy().astype(np.int32), g[p + 'rzeros'], err_msg=tag)
        else:
            assert rz is None


def test_mse_range_search_through_hip():
    """calib_algo='mse' through the host class: fp32 qparams like the reference; the searched range is a discrete
    choice decided by sums of |q - x|^2.4, so rows may differ from the reference only on near-ties (< 3 %), and rows
    with the same range must have bit-identical qparams and fake-quantized weights."""
    from llmc_amd.compression.quantization import IntegerQuantizer
    g = load_golden('mse')
    total = same_total = 0
    for ci, c in enumerate(g['cases']):
        dt, bit, sym, gran, gs = str(c).split('|')
        sym, gs = sym == 'True', (None if gs == 'None' else int(gs))
        p = f'c{ci}_'
        kw = dict(group_size=gs) if gs else {}
        q = IntegerQuantizer(int(bit), sym, gran, calib_algo='mse', **kw)
        w = dev(g[p + 'w'], dt)
        t = q.reshape_tensor(w)
        mn, mx = q.get_tensor_range(t)
        same = (host(mn).reshape(-1) == g[p + 'min']) & (host(mx).reshape(-1) == g[p + 'max'])
        total += same.size
        same_total += int(same.sum())
        _, s, z, _, _ = q.get_tensor_qparams(w)
        assert s.dtype == torch.float32
        np.testing.assert_array_equal(bits(host(s).reshape(-1))[same], bits(g[p + 'scales'])[same], err_msg=str(c))
        if not sym:
            assert z.dtype == torch.float32
            np.testing.assert_array_equal(host(z).reshape(-1)[same], g[p + 'zeros'][same], err_msg=str(c))
        fq = q.fake_quant_weight_dynamic(w)
        assert fq.dtype == TD[dt] and fq.shape == w.shape
        rows = host(fq).reshape(same.size, -1)
        ref = g[p + 'fake'].reshape(same.size, -1)
        np.testing.assert_array_equal(bits(rows[same]), bits(ref[same]), err_msg=str(c))
        codes, rs, rz = q.real_quant_weight_dynamic(w)
        assert rs.dtype == torch.float32 and codes.shape == w.shape
    assert same_total >= 0.97 * total, (same_total, total)


def test_golden_static_through_hip():
    g = load_golden('quant')
    for ci in range(int(g['n_static'])):
        p = f's{ci}_'
        bit, sym, gs, qmin, qmax = g[p + 'meta']
        wdt, sdt, zdt = [str(x) for x in g[p + 'dts']]
        q = make_quantizer(bit, sym, 'per_group', int(gs))
        w = dev(g[p + 'w'], wdt)
        s = dev(g[p + 'scales'].reshape(-1, 1), sdt)
        z = dev(g[p + 'zeros'].reshape(-1, 1), zdt) if zdt != 'none' else torch.tensor(0.0)
        args = {'scales': s, 'zeros': z, 'qmax': torch.tensor(qmax), 'qmin': torch.tensor(qmin)}
        fq = q.fake_quant_weight_static(w, dict(args))
        assert str(fq.dtype) == str(g[p + 'fake_dtype'])
        np.testing.assert_array_equal(bits(host(fq)), bits(g[p + 'fake']), err_msg=f'static {ci}')
        codes, _, _ = q.real_quant_weight_static(w, dict(args))
        np.testing.assert_array_equal(codes.cpu().numpy().astype(np.int32), g[p + 'codes'])


@pytest.mark.parametrize('dt', ['f16', 'bf16', 'f32'])
@pytest.mark.parametrize('cfg', [(4, False, 'per_group', 128), (4, True, 'per_group', 128),
                                 (8, True, 'per_channel', 0), (8, True, 'per_tensor', 0),
                                 (4, False, 'per_group', 64)])
def test_llama_shape_vs_oracle(dt, cfg):
    bit, sym, gran, gs = cfg
    R, K = 1024, 4096
    gen = torch.Generator().manual_seed(5 + bit)
    w = (torch.randn(R, K, generator=gen) * 0.02)
    w[:, ::97] *= 20
    w = w.to(TD[dt])
    wn = w.float().numpy()
    q = make_quantizer(bit, sym, gran, gs)
    qmin, qmax = Q.int_range(bit, sym)
    w2 = Q.reshape_rows(wn, gran, gs or None)
    fq_ref, s_ref, z_ref = Q.fake_quant_dynamic(w2, dt, sym, qmin, qmax)
    wd = w.cuda()
    fq = q.fake_quant_weight_dynamic(wd)
    np.testing.assert_array_equal(bits(host(fq)), bits(fq_ref.reshape(R, K)))
    codes_ref, _, _ = Q.real_quant_dynamic(w2, dt, sym, qmin, qmax)
    codes, rs, rz = q.real_quant_weight_dynamic(wd)
    np.testing.assert_array_equal(codes.cpu().numpy().astype(np.int32), codes_ref.reshape(R, K))
    np.testing.assert_array_equal(bits(host(rs).reshape(-1)), bits(s_ref.reshape(-1)))


@pytest.mark.parametrize('dt', ['bf16', 'f32'])
@pytest.mark.parametrize('sym', [False, True])
def test_groups_outside_the_plain_range_take_the_ieee_division(dt, sym):
    """The dynamic quantizer kernels divide by a hoisted reciprocal only when a group is "plain" (scale in
    [2^-40, 2^40), |w| < 2^40); groups that are not (huge, tiny-scaled, denormal, inf-free extremes) must match the
    oracle bit for bit through the IEEE-division fallback, and plain groups next to them must be unaffected."""
    R, K, gs = 8, 512, 128
    gen = torch.Generator().manual_seed(9)
    w = (torch.randn(R, K, generator=gen) * 0.02)
    w[0, :128] *= 1e15           # |w| > 2^40
    w[1, 128:256] = 3e38 * torch.sign(w[1, 128:256])   # near the top of the fp32 / bf16 range
    w[2, 256:384] *= 1e-30       # values far below 2^-40 (scale clamps to 1e-5 / qmax: plain divisor)
    w[3, :128] = 0.0
    w[4, 384:] *= 1e-42 if dt == 'f32' else 1e-38       # denormals (fp32) / near the smallest normals
    w = w.to(TD[dt])
    wn = w.float().numpy()
    q = make_quantizer(4, sym, 'per_group', gs)
    qmin, qmax = Q.int_range(4, sym)
    w2 = Q.reshape_rows(wn, 'per_group', gs)
    fq_ref, s_ref, z_ref = Q.fake_quant_dynamic(w2, dt, sym, qmin, qmax)
    wd = w.cuda()
    fq = q.fake_quant_weight_dynamic(wd)
    np.testing.assert_array_equal(bits(host(fq)), bits(fq_ref.reshape(R, K)))
    codes_ref, _, _ = Q.real_quant_dynamic(w2, dt, sym, qmin, qmax)
    codes, rs, rz = q.real_quant_weight_dynamic(wd)
    np.testing.assert_array_equal(codes.cpu().numpy().astype(np.int32), codes_ref.reshape(R, K))
    np.testing.assert_array_equal(bits(host(rs).reshape(-1)), bits(s_ref.reshape(-1)))


def test_ragged_and_tiny_shapes():
    # rows not a multiple of the wave's rows-per-wave, g not a multiple of the 16-B vector
    for (R, K, gran, gs) in [(3, 130, 'per_channel', 0), (1, 8, 'per_channel', 0), (7, 96, 'per_group', 32),
                             (5, 70000, 'per_channel', 0), (1, 200000, 'per_tensor', 0)]:
        gen = torch.Generator().manual_seed(R * 1000 + K)
        w = torch.randn(R, K, generator=gen).half()
        q = make_quantizer(4, False if gran != 'per_tensor' else True, gran, gs)
        sym = q.sym
        qmin, qmax = Q.int_range(4, sym)
        w2 = Q.reshape_rows(w.float().numpy(), gran, gs or None)
        fq_ref, _, _ = Q.fake_quant_dynamic(w2, 'f16', sym, qmin, qmax)
        fq = q.fake_quant_weight_dynamic(w.cuda())
        np.testing.assert_array_equal(bits(host(fq)), bits(fq_ref.reshape(R, K)), err_msg=f'{R}x{K} {gran}')


def test_pack_lsb_vs_golden_and_full_size():
    from llmc_amd.compression.quantization import pack_lsb
    g = load_golden('pack')
    for ci in range(int(g['n_vllm'])):
        codes = torch.from_numpy(g[f'v{ci}_codes']).cuda()
        bit = int(g[f'v{ci}_bit'])
        if bit == 8:
            codes = codes.to(torch.int8)
        packed = pack_lsb(codes, bit)
        np.testing.assert_array_equal(packed.cpu().numpy(), g[f'v{ci}_packed'])
    codes = torch.randint(-8, 8, (4096, 4096), dtype=torch.int32)
    packed = pack_lsb(codes.cuda(), 4)
    np.testing.assert_array_equal(packed.cpu().numpy(), Q.pack_lsb(codes.numpy(), 4))


def test_cpu_tensor_is_refused():
    from llmc_amd import _ffi
    q = make_quantizer(4, True, 'per_group', 128)
    with pytest.raises(_ffi.LlmcHipError):
        q.fake_quant_weight_dynamic(torch.randn(4, 128).half())


def test_per_tensor_asymmetric_qparams_are_fp32_like_the_reference():
    """per_tensor + asymmetric — 0-dim min/max against the 0-dim fp32 (qmax - qmin) promote scales / zeros
    to fp32 (quant.py:132-136,555-556); fake-quant output stays in the weight dtype, codes match bit for bit."""
    from conftest import load_golden
    from llmc_amd.compression.quantization import IntegerQuantizer
    TDm = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}
    g = load_golden('quant_pt')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        bit = int(g[p + 'meta'][0])
        dt = TDm[str(g[p + 'dt'])]
        q = IntegerQuantizer(bit, False, 'per_tensor')
        w = torch.from_numpy(g[p + 'w']).to(dt).cuda()
        _, s, z, _, _ = q.get_tensor_qparams(w)
        assert str(s.dtype) == str(g[p + 'scales_dtype']) and str(z.dtype) == str(g[p + 'zeros_dtype']), ci
        assert s.dim() == 0 and z.dim() == 0
        np.testing.assert_array_equal(s.reshape(-1).cpu().numpy().view(np.uint32), g[p + 'scales'].view(np.uint32))
        np.testing.assert_array_equal(z.reshape(-1).cpu().numpy(), g[p + 'zeros'])
        fq = q.fake_quant_weight_dynamic(w)
        assert str(fq.dtype) == str(g[p + 'fake_dtype'])
        np.testing.assert_array_equal(fq.float().cpu().numpy().view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(ci))
        codes, rs, rz = q.real_quant_weight_dynamic(w)
        assert str(codes.dtype) == str(g[p + 'codes_dtype'])
        np.testing.assert_array_equal(codes.cpu().numpy().astype(np.int32), g[p + 'codes'], err_msg=str(ci))
        np.testing.assert_array_equal(rs.float().reshape(-1).cpu().numpy().view(np.uint32), g[p + 'rscales'].view(np.uint32))
        np.testing.assert_array_equal(rz.reshape(-1).cpu().numpy().astype(np.int32), g[p + 'rzeros'])


def test_static_hist_range_matches_reference_golden():
    """llmc_histc (torch.histc's binning) bit-exact against the oracle on awkward ranges, and the whole static_hist
    calibration (data pass on the GPU + host search) equal to the reference's range on the goldens."""
    from llmc_amd.compression.quantization.hist_range import HistRange, static_hist_range
    from oracle import hist_ref as Hs
    g = load_golden('hist')
    rs = np.random.RandomState(0)
    for dt, n in ((torch.bfloat16, 1000003), (torch.float16, 4099), (torch.float32, 77777)):
        x = torch.from_numpy((rs.randn(n) * 3).astype(np.float32)).to(dt).cuda()
        xf = x.float().cpu().numpy()
        for lo, hi in ((float(xf.min()), float(xf.max())), (-1.7, 2.9), (0.25, 0.25)):
            got = HistRange(2048)._histc(x, lo, hi)
            np.testing.assert_array_equal(got, Hs.histc(xf, 2048, lo, hi))
    for name in sorted({k.split('/')[0] for k in g.files}):
        p = name + '/'
        xs = [torch.from_numpy(s).to(torch.bfloat16).cuda() for s in g[p + 'x']]
        h = HistRange(2048, 16, 256)
        for s in xs:
            h.add(s)
        np.testing.assert_array_equal(h.hist, g[p + 'hist'], err_msg=name)
        lo, hi = static_hist_range(xs)
        assert np.float32(lo) == g[p + 'new_min'] and np.float32(hi) == g[p + 'new_max'], (name, lo, hi)
        # get_qparams on the searched range (quant.py:545-553): the scale the reference registers
        s_ref = float(g[p + 'scale'])
        s = max(abs(lo), abs(hi)) / 127.0
        assert abs(np.float32(s) - np.float32(s_ref)) <= 1e-7 * abs(s_ref), name


@pytest.mark.parametrize('algo', ['static_hist', 'static_minmax', 'static_moving_minmax'])
def test_register_act_qparams_all_static_calibrations(algo):
    """BaseBlockwiseQuantization.register_act_qparams (base_blockwise_quantization.py:567-588) for the three calibrations
    get_batch_tensors_qparams accepts (quant.py:561-574); static_hist against the reference's scale (goldens), the others
    against their definitions (mean of per-sample extrema; exponential moving average with alpha = 0.01)."""
    from llmc_amd.compression.quantization.base_blockwise_quantization import BaseBlockwiseQuantization
    from llmc_amd.compression.quantization.quant import IntegerQuantizer
    g = load_golden('hist')
    p = 'growing/'
    xs = [torch.from_numpy(s).to(torch.bfloat16).cuda().unsqueeze(0) for s in g[p + 'x']]
    obj = BaseBlockwiseQuantization.__new__(BaseBlockwiseQuantization)
    obj.aquantizer = IntegerQuantizer(8, True, 'per_tensor', calib_algo=algo)
    layer = torch.nn.Linear(4, 4).cuda()
    obj.register_act_qparams({'fc': layer}, list(xs))
    s = float(layer.buf_act_scales_0)
    mx = [float(x.max()) for x in xs]
    mn = [float(x.min()) for x in xs]
    if algo == 'static_hist':
        assert abs(s - float(g[p + 'scale'])) <= 1e-6 * float(g[p + 'scale'])
    elif algo == 'static_minmax':
        assert abs(s - max(abs(np.mean(mx)), abs(np.mean(mn))) / 127) <= 1e-3 * s
    else:
        a = b = None
        for lo, hi in zip(mn, mx):
            a, b = (lo, hi) if a is None else (a + 0.01 * (lo - a), b + 0.01 * (hi - b))
        assert abs(s - max(abs(a), abs(b)) / 127) <= 1e-2 * s
    assert float(layer.buf_act_qmax_0) == 127 and float(layer.buf_act_zeros_0) == 0


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16, torch.float32])
def test_sample_minmax_equals_torch_min_max_per_sample(dt):
    """llmc_minmax_samples (the data pass of static_minmax / static_moving_minmax / the histogram observer's first pass,
    quant.py:253-263): every sample's min / max exactly what `sample.min()`, `sample.max()` give, for separately allocated
    samples of different lengths (ragged tails, one shorter than a vector), more samples than one launch takes, NaN."""
    from llmc_amd.compression.quantization.hist_range import sample_minmax
    gen = torch.Generator().manual_seed(3)
    lens = [1, 7, 8, 4099, 65536, 65537, 200003] + [1000 + 13 * i for i in range(170)]
    xs = [(torch.randn(n, generator=gen) * (1 + i % 5)).to(dt).cuda() for i, n in enumerate(lens)]
    xs[3][17] = float('nan')
    xs[5] = xs[5].reshape(1, -1, 1)                  # any shape: the sample is its elements
    mn, mx = sample_minmax(xs)
    assert mn.dtype == torch.float32 and mn.shape == (len(xs),)
    for i, x in enumerate(xs):
        a, b = x.min().float(), x.max().float()
        if i == 3:
            assert torch.isnan(mn[i]) and torch.isnan(mx[i]) and torch.isnan(a)      # torch propagates the NaN too
        else:
            assert float(mn[i]) == float(a) and float(mx[i]) == float(b), i
