"""The reference-NAMED per-layer methods of GPTQ / SpQR / Awq (gptq.py:58-244, 333-409; spqr.py:185-254, 323-355; awq.py:40-46,
147-164) called the way the reference's own layer_transform calls them, on the golden inputs the reference produced
(tests/golden/gptq.npz, gptq_more.npz, gptq_owq.npz, spqr.npz, awq.npz): same outputs and the same in-place effects."""
import math
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TD = {'f16': torch.float16, 'bf16': torch.bfloat16, 'torch.float16': torch.float16, 'torch.bfloat16': torch.bfloat16,
      'torch.float32': torch.float32}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def host(t):
    return t.detach().float().cpu().numpy()


class _Acc:
    """stands in for HessianAccumulator: the golden Hessian, already complete"""
    def __init__(self, H):
        self.H = H
        self.nsamples = 1


def _gptq(g, p, owq_n_out=0):
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.gptq import GPTQ
    meta = g[p + 'meta']
    if owq_n_out:                   # gptq_owq.npz: [bit, sym, gs, R, K, n_out, qmin, qmax]
        bit, sym, gs, R, K = [int(v) for v in meta[:5]]
        actorder = static_groups = 0
    else:                           # gptq.npz: [bit, sym, gs, actorder, static_groups, R, K, qmin, qmax]
        bit, sym, gs, actorder, static_groups, R, K = [int(v) for v in meta[:7]]
    gran = str(g[p + 'gran'])
    wq = IntegerQuantizer(bit, bool(sym), gran, **({'group_size': gs} if gran == 'per_group' else {}))
    a = object.__new__(GPTQ)
    a.wquantizer, a.actorder, a.static_groups = wq, bool(actorder), bool(static_groups)
    a.owq, a.percdamp, a.blocksize = bool(owq_n_out), 0.01, 128
    if a.owq:
        a.actorder, a.static_groups = False, False
    a.model_dtype = TD[str(g[p + 'dt'])]
    a.layers_cache, a._groups, a._group_of = {}, {}, {}
    layer = torch.nn.Linear(K, R, bias=False).cuda()
    layer.weight.data = torch.from_numpy(g[p + 'W0']).to(a.model_dtype).cuda()
    # collect_block_qparams (base_blockwise_quantization.py:338-365): RTN qparams of the original weights
    _, s, z, qmax, qmin = wq.get_tensor_qparams(layer.weight.data)
    layer.register_buffer('buf_scales', s)
    layer.register_buffer('buf_zeros', z if torch.is_tensor(z) and z.dim() > 0 else torch.tensor(0.0))
    layer.register_buffer('buf_qmax', qmax)
    layer.register_buffer('buf_qmin', qmin)
    a.layers_cache['l'] = {'columns': K, 'nsamples': 1}
    if owq_n_out:
        a.n_out_dict = {'l': owq_n_out}
    return a, layer, dict(R=R, K=K, gs=gs, sym=bool(sym), dynamic=gran == 'per_group' and not a.static_groups)


def _bind_hessian(a, H):
    a._groups[1] = {'acc': _Acc(H), 'pass': None, 'passes': 1, 'members': ['l']}
    a._group_of['l'] = 1
    a.layers_cache['l'].update({'acc': a._groups[1]['acc'], 'joined': 1})


def test_gptq_per_layer_flow_by_reference_names_on_the_goldens_with_a_hessian():
    """gptq.npz carries H: hessian_sorting -> process_hessian_and_weights -> weight_transform -> the layer's end state."""
    g = load_golden('gptq')
    checked = 0
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        if (p + 'H') not in g.files:
            continue
        checked += 1
        a, layer, c = _gptq(g, p)
        H = torch.from_numpy(g[p + 'H'].copy()).cuda()
        _bind_hessian(a, H)
        a.initialize_qparams_and_prepare_weights(layer, 'l')
        assert a.columns == c['K'] and a.n_nonout == c['K'] and a.qparams == {}
        if a.actorder:
            d = torch.diagonal(H).cpu().numpy()
            np.testing.assert_array_equal(d[a.perm.cpu().numpy()], d[g[p + 'perm']], err_msg=name)   # equal up to exact ties
            a.perm = torch.from_numpy(g[p + 'perm']).cuda()
        W, U = a.process_hessian_and_weights(layer, 'l')
        np.testing.assert_array_equal(bits(host(W)), bits(g[p + 'Wp']), err_msg=name)
        Ug = g[p + 'U']
        assert np.abs(host(U) - Ug).max() / np.abs(Ug).max() < 2e-4, name      # another LAPACK's factor
        assert torch.equal(torch.tril(U, -1), torch.zeros_like(U))
        if a.actorder:
            assert torch.equal(layer.buf_perm, a.perm) and torch.equal(layer.buf_invperm, torch.argsort(a.perm))
        if not c['dynamic']:                          # ready() was False: the RTN qparams were read from the layer
            assert (len(a.groups) == c['K'] // c['gs']) if c['gs'] else ('scale' in a.qparams)
        # the loop on the REFERENCE's factor: bit-exact results and in-place effects
        Wg, Ugd = torch.from_numpy(g[p + 'Wp'].copy()).cuda(), torch.from_numpy(Ug.copy()).cuda()
        a.update_layer_with_transformed_weights(layer, Wg, Ugd, 'l')
        np.testing.assert_array_equal(bits(host(layer.weight.data)), bits(g[p + 'final_w']), err_msg=name)
        assert layer.weight.dtype == torch.float32
        if c['dynamic']:
            np.testing.assert_array_equal(bits(host(layer.buf_scales).ravel()), bits(g[p + 'buf_scales'].ravel()), err_msg=name)
            if not c['sym']:
                np.testing.assert_array_equal(host(layer.buf_zeros).ravel(), g[p + 'buf_zeros'].ravel(), err_msg=name)
            assert layer.buf_scales.shape == (c['R'] * c['K'] // c['gs'], 1) and layer.buf_scales.dtype == torch.float32
    assert checked >= 2


def test_gptq_weight_transform_by_name_mutates_losses_tmp_and_groups_like_the_reference():
    g = load_golden('gptq+more')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        a, layer, c = _gptq(g, p)
        a.columns, a.n_out, a.n_nonout, a.qparams = c['K'], 0, c['K'], {}
        if g[p + 'perm'].size:
            a.perm = torch.from_numpy(g[p + 'perm']).cuda()
        if not c['dynamic']:
            # static groups / per_channel read the layer's qparams exactly as process_hessian_and_weights leaves them
            sdt = TD[str(g[p + 'buf_scales_dtype'])]
            layer.buf_scales = torch.from_numpy(g[p + 'buf_scales']).to(sdt).reshape(-1, 1).cuda()
            if g[p + 'buf_zeros'].size:
                layer.buf_zeros = torch.from_numpy(g[p + 'buf_zeros']).to(sdt).reshape(-1, 1).cuda()
            if c['gs']:
                a.groups = []
                a.search_group_qparams(layer)
                assert len(a.groups) == c['K'] // c['gs'] and a.groups[0]['scale'].shape == (c['R'], 1)
                np.testing.assert_array_equal(host(a.merge_qparams([q['scale'] for q in a.groups])), host(layer.buf_scales))
            else:
                a.search_layer_qparams(layer)
        W = torch.from_numpy(g[p + 'Wp'].copy()).cuda()
        U = torch.from_numpy(g[p + 'U'].copy()).cuda()
        Losses, tmp = torch.zeros_like(W), torch.zeros_like(W)
        ret = a.weight_transform(W, U, Losses, tmp)
        assert ret is None
        np.testing.assert_array_equal(bits(host(tmp)), bits(g[p + 'tmp']), err_msg=name)
        np.testing.assert_array_equal(bits(host(Losses)), bits(g[p + 'losses']), err_msg=name)
        if c['dynamic']:
            s = torch.cat([q['scale'] for q in a.groups], 1)
            np.testing.assert_array_equal(bits(host(s)), bits(g[p + 'g_scales']), err_msg=name)
            if not c['sym']:
                np.testing.assert_array_equal(host(torch.cat([q['zero'] for q in a.groups], 1)), g[p + 'g_zeros'], err_msg=name)
            assert bool(a.ready())
            # search_column_qparams by name: the first group's qparams are min/max of the untouched first columns
            a2, _, _ = _gptq(g, p)
            a2.qparams, a2.groups = {}, [None] * (c['K'] // c['gs'])
            a2.search_column_qparams(torch.from_numpy(g[p + 'Wp'][:, :c['gs']].copy()).cuda(), 0)
            np.testing.assert_array_equal(bits(host(a2.groups[0]['scale'])), bits(g[p + 'g_scales'][:, :1]), err_msg=name)


def test_gptq_owq_flow_by_reference_names():
    g = load_golden('gptq_owq')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        n_out = int(g[p + 'meta'][5])
        a, layer, c = _gptq(g, p, owq_n_out=n_out)
        # hessian_sorting's OWQ branch from the golden diagonal
        K = c['K']
        H = torch.eye(K, device='cuda') * torch.from_numpy(g[p + 'Hdiag']).cuda()
        _bind_hessian(a, H)
        a.initialize_qparams_and_prepare_weights(layer, 'l')
        assert a.n_out == n_out and a.n_nonout == K - n_out
        np.testing.assert_array_equal(a.perm.cpu().numpy(), g[p + 'perm'], err_msg=name)
        a.invperm = torch.argsort(a.perm)
        if c['gs']:
            a.groups = []
            a.search_group_qparams(layer)          # the groups OWQ never visits keep these RTN qparams (gptq.py:380-395)
        else:
            _, s, z, _, _ = a.wquantizer.get_tensor_qparams(torch.from_numpy(g[p + 'Wp'][:, :K - n_out].copy()).cuda())
            a.qparams = {'scale': s, 'zero': z}
        W = torch.from_numpy(g[p + 'Wp'].copy()).cuda()
        U = torch.from_numpy(g[p + 'U'].copy()).cuda()
        Losses, tmp = torch.zeros_like(W), torch.zeros_like(W)
        a.weight_transform(W, U, Losses, tmp)
        np.testing.assert_array_equal(bits(host(tmp)), bits(g[p + 'tmp']), err_msg=name)
        np.testing.assert_array_equal(bits(host(Losses)), bits(g[p + 'losses']), err_msg=name)
        # the floating-point outlier columns of W ended with every block's feedback (what gptq.py:187 copies into tmp)
        np.testing.assert_array_equal(bits(host(W[:, K - n_out:])), bits(g[p + 'W_after'][:, K - n_out:]), err_msg=name)


@pytest.mark.parametrize('name', ['g16_act_thr02', 'g32_noact_thr01', 'g16_act_inf', 'g64_act_simplified'])
def test_spqr_weight_transform_and_group_qparams_by_reference_names(name):
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.spqr import SpQR, SpqrConfig
    g = load_golden('spqr')
    p = name + '/'
    bit, gs, act, R, K, simp = [int(v) for v in g[p + 'cfg']]
    a = object.__new__(SpQR)
    a.wquantizer = IntegerQuantizer(bit, False, 'per_group', group_size=gs, round_zp=False)
    a.scale_quantizer = IntegerQuantizer(3, False, 'per_group', group_size=16, round_zp=False)
    a.zero_quantizer = IntegerQuantizer(3, False, 'per_group', group_size=16, round_zp=False)
    a.relative_threshold, a.simplified_outliers, a.blocksize, a.columns = float(g[p + 'rel_threshold']), bool(simp), 128, K
    a.scfg = SpqrConfig(bit=bit, group_size=gs, actorder=bool(act), percdamp=float(g[p + 'percdamp']),
                        relative_threshold=a.relative_threshold, simplified_outliers=bool(simp))
    W = torch.from_numpy(g[p + 'Wp'].copy()).cuda()
    U = torch.from_numpy(g[p + 'U'].copy()).cuda()
    Losses, tmp, mask = torch.zeros_like(W), torch.zeros_like(W), torch.zeros_like(W, dtype=torch.bool)
    a.weight_transform(W, U, Losses, tmp, mask)
    if not math.isinf(a.relative_threshold):
        assert abs(a.last_threshold - float(g[p + 'threshold'])) <= 1e-5 * abs(float(g[p + 'threshold']))
    if abs(a.last_threshold - float(g[p + 'threshold'])) == 0 or math.isinf(a.relative_threshold):
        np.testing.assert_array_equal(mask.cpu().numpy().astype(np.uint8), g[p + 'mask'])
        np.testing.assert_array_equal(host(tmp), g[p + 'tmp'])
        np.testing.assert_array_equal(host(Losses), g[p + 'losses'])
    layer = torch.nn.Linear(K, R, bias=False).cuda()
    a.set_model_qparams(layer)
    if abs(a.last_threshold - float(g[p + 'threshold'])) == 0 or math.isinf(a.relative_threshold):
        np.testing.assert_array_equal(host(layer.buf_scales), g[p + 'buf_scales'])
        np.testing.assert_array_equal(host(layer.buf_zeros), g[p + 'buf_zeros'])
    assert float(layer.buf_qmax) == 2 ** bit - 1 and float(layer.buf_qmin) == 0
    if simp or math.isinf(a.relative_threshold):
        # without outlier detection the first group's qparams come from the untouched first columns: get_group_qparams by
        # name (the quantizers' kernels) equals what the column loop's kernel produced
        b = object.__new__(SpQR)
        b.__dict__.update(a.__dict__)
        b.groups, b.qparams = [None] * (K // gs), {}
        b.get_group_qparams(torch.from_numpy(g[p + 'Wp'][:, :gs].copy()).cuda(), 0)
        np.testing.assert_array_equal(host(b.groups[0]['scales']), g[p + 'buf_scales'].reshape(R, K // gs)[:, :1])
        np.testing.assert_array_equal(host(b.groups[0]['zeros']), g[p + 'buf_zeros'].reshape(R, K // gs)[:, :1])


def test_awq_fake_quantize_weight_and_scaling_weight_by_reference_names():
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.awq import Awq
    g = load_golden('awq+more')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, nl, K, bit = [int(v) for v in g[p + 'meta']]
        dt = TD[str(g[p + 'dt'])]
        a = object.__new__(Awq)
        a.wquantizer = (IntegerQuantizer(bit, bool(sym), 'per_group', group_size=gs) if gs
                        else IntegerQuantizer(bit, bool(sym), 'per_channel'))
        s = torch.from_numpy(g[p + 'scales_r035']).to(dt).cuda()
        r0 = 0
        for i in range(nl):
            w = torch.from_numpy(g[p + f'w{i}']).to(dt).cuda()
            fc = torch.nn.Linear(K, w.shape[0], bias=False).cuda()
            fc.weight.data = w.clone()
            out = a.fake_quantize_weight(fc, s, False, f'l{i}')
            assert out is fc.weight and fc.weight.dtype == dt
            np.testing.assert_array_equal(bits(host(fc.weight.data)), bits(g[p + 'wq_r035'][r0:r0 + w.shape[0]]), err_msg=name)
            r0 += w.shape[0]
            w2 = w.clone()
            ret = a.scaling_weight(w2, s, False)
            assert ret.data_ptr() == w2.data_ptr()                      # in place, like w.mul_(scales.view(1, -1))
            np.testing.assert_array_equal(bits(host(w2)), bits((w.float() * s.float().view(1, -1)).to(dt).float().cpu().numpy()),
                                          err_msg=name)


def test_static_activation_ranges_by_reference_names():
    """BaseQuantizer.get_batch_tensors_qparams / get_static_minmax_range / get_minmax_stats (quant.py:221-263, 561-586) on a list
    of separately allocated samples against plain torch."""
    from llmc_amd.compression.quantization import IntegerQuantizer
    gen = torch.Generator().manual_seed(3)
    samples = [(torch.randn(1, 70 + 3 * i, 256, generator=gen) * (1 + i)).to(torch.bfloat16).cuda() for i in range(5)]
    q = IntegerQuantizer(8, True, 'per_tensor', calib_algo='static_minmax')
    mn = torch.stack([t.float().min() for t in samples]).mean()
    mx = torch.stack([t.float().max() for t in samples]).mean()
    lo, hi = q.get_static_minmax_range(list(samples))
    assert len(lo) == 1 and float(lo[0]) == float(mn) and float(hi[0]) == float(mx)
    sl, zl, qmin_l, qmax_l = q.get_batch_tensors_qparams(list(samples))
    want = torch.max(mx.abs(), mn.abs()).clamp(min=1e-5) / 127
    assert len(sl) == 1 and float(sl[0]) == float(want) and float(zl[0]) == 0.0 and float(qmax_l[0]) == 127
    one = torch.cat([t for t in samples[:1]] * 3, 0)                  # the bs = -1 form: one tensor, samples along dim 0
    lo1, hi1 = q.get_static_minmax_range([one])
    assert float(lo1[0]) == float(samples[0].float().min()) and float(hi1[0]) == float(samples[0].float().max())
    r = IntegerQuantizer(4, False, 'per_channel', calib_algo='mse').get_mse_range(torch.randn(16, 256, generator=gen).cuda())
    assert r[0].shape == (16, 1) and bool((r[0] <= 0).all()) and bool((r[1] >= 0).all())
    with pytest.raises(ValueError):
        IntegerQuantizer(8, True, 'per_tensor').get_batch_tensors_qparams(list(samples))      # 'minmax': quant.py:573-574


def test_float_quantizer_quant_dequant_by_reference_names():
    from llmc_amd.compression.quantization import FloatQuantizer
    q = FloatQuantizer('e4m3', True, 'per_channel', use_qtorch=True)
    gen = torch.Generator().manual_seed(5)
    w = (torch.randn(64, 256, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    t, s, z, qmax, qmin = q.get_tensor_qparams(w)
    codes = q.quant(t, s, z, qmax, qmin)
    assert codes.dtype == torch.float32 and float(codes.abs().max()) <= 240.0          # qtorch saturates e4m3 at 240
    assert torch.equal(codes, codes.to(torch.float8_e4m3fn).float())                   # on the 8-bit grid
    fq = q.quant_dequant(t, s, z, qmax, qmin)
    want = q.fake_quant_weight_dynamic(w)
    assert torch.equal(fq.to(w.dtype), want)
