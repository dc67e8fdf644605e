"""Host-side mirror of llmc's operator surface: registry protocol, signatures, config plumbing (CPU, no compute)."""
import inspect

import pytest
import torch

import llmc_amd.compression.quantization as Q
from llmc_amd.utils.registry_factory import ALGO_REGISTRY, Register


def test_registry_protocol_like_llmc():
    assert {'GPTQ', 'Awq', 'RTN'} <= set(ALGO_REGISTRY.keys())
    r = Register()

    @r
    class A:
        pass

    @r('other')
    class B:
        pass
    assert r['A'] is A and r['other'] is B and 'A' in r
    with pytest.raises(Exception):
        r.register(A)
    with pytest.raises(Exception):
        r.register('k')(3)          # values must be callable


def test_operator_signatures_match_the_reference_surface():
    # SURVEY.md §8b "Python operator signatures (must not change)"
    sig = lambda f: list(inspect.signature(f).parameters)  # noqa: E731
    assert sig(Q.GPTQ.__init__)[:6] == ['self', 'model', 'quant_config', 'input', 'padding_mask', 'config']
    assert sig(Q.Awq.__init__) == ['self', 'model', 'quant_config', 'input', 'padding_mask', 'config']
    assert sig(Q.RTN.__init__) == ['self', 'model', 'quant_config', 'input', 'padding_mask', 'config']
    assert sig(Q.GPTQ.add_batch) == ['self', 'layer', 'name', 'inp', 'out']
    assert sig(Q.GPTQ.layer_transform) == ['self', 'layer', 'name']
    assert sig(Q.GPTQ.subset_transform) == ['self', 'subset', 'input_feat', 'subset_kwargs']
    assert sig(Q.GPTQ.cache_input_hook) == ['self', 'm', 'inp', 'out', 'name', 'feat_dict']
    assert sig(Q.GPTQ.w_qdq) == ['self', 'module', 'wquantizer']
    assert sig(Q.GPTQ.w_q) == ['self', 'module', 'wquantizer']
    assert sig(Q.BaseBlockwiseQuantization.a_qdq) == ['self', 'act', 'module', 'aquantizer', 'input_index']
    assert sig(Q.BaseBlockwiseQuantization.block_transform) == ['self', 'block', 'input_feat', 'block_kwargs']
    assert sig(Q.Awq.search_scale_subset) == ['self', 'prev_op', 'layers_dict', 'input', 'inspect_module', 'is_gqa',
                                              'subset_kwargs']
    assert sig(Q.IntegerQuantizer.get_tensor_qparams) == ['self', 'tensor', 'args']
    assert sig(Q.IntegerQuantizer.quant_dequant)[:6] == ['self', 'tensor', 'scales', 'zeros', 'qmax', 'qmin']
    for name in ('fake_quant_weight_dynamic', 'fake_quant_weight_static', 'real_quant_weight_dynamic',
                 'real_quant_weight_static', 'fake_quant_act_dynamic', 'fake_quant_act_static'):
        assert hasattr(Q.IntegerQuantizer, name) and hasattr(Q.FloatQuantizer, name)
    for cls in (Q.FakeQuantLinear, Q.EffcientFakeQuantLinear, Q.VllmRealQuantLinear, Q.AutoawqRealQuantLinear):
        assert sig(cls.new)[0] == 'module'


def test_quantizer_ranges_and_views_without_gpu():
    q = Q.IntegerQuantizer(4, False, 'per_group', group_size=128)
    assert float(q.qmin) == 0.0 and int(q.qmax) == 15 and q.qmin.dtype == torch.float32 and q.qmax.dtype == torch.int64
    q = Q.IntegerQuantizer(8, True, 'per_channel')
    assert int(q.qmin) == -128 and int(q.qmax) == 127
    q = Q.IntegerQuantizer(4, True, 'per_group', group_size=64)
    t = torch.zeros(6, 256)
    assert q.reshape_tensor(t).shape == (24, 64)
    assert q.restore_tensor(q.reshape_tensor(t), t.shape).shape == t.shape
    with pytest.raises(ValueError):
        q.reshape_tensor(torch.zeros(2, 100))
    qm = Q.IntegerQuantizer(4, True, 'per_group', group_size=64, calib_algo='mse')
    assert (qm.maxshrink, qm.mse_grid, qm.mse_b_num) == (0.8, 100, 1)
    with pytest.raises(NotImplementedError):
        Q.IntegerQuantizer(4, True, 'per_group', group_size=64, calib_algo='hist')
    f = Q.FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True)
    assert float(f.qmax) == 448.0


def test_out_of_scope_config_is_rejected_loudly():
    from toy_model import ToyModel, calib_input
    model = ToyModel()
    cfg = {'weight': {'bit': 4, 'symmetric': True, 'granularity': 'per_group', 'group_size': 128},
           'kvcache': {'method': 'Naive'}}
    with pytest.raises(NotImplementedError):
        Q.RTN(model, cfg, calib_input(model), None, {})


def test_gptq_hessian_sharing_needs_the_same_input_tensor(monkeypatch):
    """layers of one subset share a Hessian only while their hooked inputs are the very same
    tensor. MoE-style subset: two experts see different (routed) tokens, the router sees all of them."""
    import types

    import llmc_amd.compression.quantization.gptq as gq

    class FakeAcc:
        def __init__(self, K, dev, defer=True, exact_diag=True):
            self.K, self.nsamples, self.fed = K, 0, []
            self.H = torch.zeros(K, K)

        def add(self, inp):
            self.fed.append(inp)
            self.nsamples += inp.shape[0]
    monkeypatch.setattr(gq, 'HessianAccumulator', FakeAcc)
    g = gq.GPTQ.__new__(gq.GPTQ)
    g.layers_cache, g._groups, g._group_of = {}, {}, {}
    lin = lambda: torch.nn.Linear(8, 4, bias=False)  # noqa: E731
    layers = {'experts.0.w1': lin(), 'experts.0.w3': lin(), 'experts.1.w1': lin(), 'experts.1.w3': lin(), 'gate': lin()}
    g.subset_init({'layers': layers})
    assert len(g._groups) == 1                                     # same in_features: one group to start with
    for step in range(2):
        x_all = torch.randn(1, 6, 8)
        x0, x1 = x_all[:, :4], x_all[:, 4:]                         # routed tokens: different views
        for n, x in (('gate', x_all), ('experts.0.w1', x0), ('experts.0.w3', x0), ('experts.1.w1', x1),
                     ('experts.1.w3', x1)):
            g.add_batch(layers[n], n, x, None)
    gids = {n: g._group_of[n] for n in layers}
    assert len(set(gids.values())) == 5
    # the router keeps the original group; each expert's w1 left it; w3 saw a different tensor than the router too
    assert gids['gate'] != gids['experts.0.w1'] != gids['experts.1.w1']
    for n in layers:
        acc = g.layers_cache[n]['acc']
        assert len(acc.fed) == 2 and acc.nsamples == 2, n           # every layer's own Hessian saw both batches
    assert g.layers_cache['gate']['acc'].fed[0].shape[1] == 6 and g.layers_cache['experts.1.w3']['acc'].fed[0].shape[1] == 2
    # Llama-style subset: the same tensor object -> one accumulator, fed once per call
    g2 = gq.GPTQ.__new__(gq.GPTQ)
    g2.layers_cache, g2._groups, g2._group_of = {}, {}, {}
    qkv = {'q': lin(), 'k': lin(), 'v': lin()}
    g2.subset_init({'layers': qkv})
    for step in range(3):
        h = torch.randn(1, 5, 8)
        for n in qkv:
            g2.add_batch(qkv[n], n, h, None)
    assert len(g2._groups) == 1
    acc = g2.layers_cache['q']['acc']
    assert acc is g2.layers_cache['v']['acc'] and len(acc.fed) == 3 and g2.layers_cache['k']['nsamples'] == 3
    # a member that changes its mind later cannot be repaired: loud failure
    with pytest.raises(RuntimeError):
        h = torch.randn(1, 5, 8)
        g2.add_batch(qkv['q'], 'q', h, None)
        g2.add_batch(qkv['k'], 'k', h.clone(), None)


def test_gptq_hessian_sharing_with_skipped_experts_and_pass_ids(monkeypatch):
    """an expert that receives no token is skipped by the HF / DeepSeek forward, so the members of a
    subset are called a different number of times. Sharing is keyed on the forward pass (block_forward's counter, or a
    member being called again), the feeder's tensor stays referenced during the pass, and a member whose first call
    comes after the group was fed leaves it — it must never inherit (or feed) the router's Hessian."""
    import llmc_amd.compression.quantization.gptq as gq

    class FakeAcc:
        def __init__(self, K, dev, defer=True, exact_diag=True):
            self.K, self.nsamples, self.fed = K, 0, []
            self.H = torch.zeros(K, K)

        def add(self, inp):
            self.fed.append(tuple(inp.shape))
            self.nsamples += inp.shape[0]
    monkeypatch.setattr(gq, 'HessianAccumulator', FakeAcc)
    lin = lambda: torch.nn.Linear(8, 4, bias=False)  # noqa: E731
    for with_pass_id in (True, False):
        for e1_first in (False, True):
            g = gq.GPTQ.__new__(gq.GPTQ)
            g.layers_cache, g._groups, g._group_of = {}, {}, {}
            layers = {'gate': lin(), 'e0.w1': lin(), 'e0.w3': lin(), 'e1.w1': lin(), 'e1.w3': lin(), 'e2.w1': lin()}
            g.subset_init({'layers': layers})
            # sample 1: the router sends every token to expert 0; sample 2: tokens for experts 0 and 1; expert 2 never runs
            x1 = torch.randn(1, 6, 8)
            if with_pass_id:
                g._fwd_pass = 1
            for n, x in (('gate', x1), ('e0.w1', x1[:, :6]), ('e0.w3', x1[:, :6])):
                g.add_batch(layers[n], n, x[:, :4] if n != 'gate' else x, None)
            x2 = torch.randn(1, 6, 8)
            # the allocator may hand sample 1's routed buffer to another expert: same pointer / shape / stride
            r0 = x2[:, :4]
            r1 = x2[:, 4:]
            if with_pass_id:
                g._fwd_pass = 2
            calls = [('gate', x2), ('e0.w1', r0), ('e0.w3', r0), ('e1.w1', r1), ('e1.w3', r1)]
            if e1_first and with_pass_id:
                calls = [calls[3], calls[4], calls[0], calls[1], calls[2]]
            for n, x in calls:
                g.add_batch(layers[n], n, x, None)
            fed = {n: g.layers_cache[n]['acc'].fed for n in layers}
            assert fed['gate'] == [(1, 6, 8), (1, 6, 8)], (with_pass_id, e1_first, fed['gate'])
            assert fed['e0.w1'] == [(1, 4, 8), (1, 4, 8)] and fed['e0.w3'] == [(1, 4, 8), (1, 4, 8)]
            assert fed['e1.w1'] == [(1, 2, 8)] and fed['e1.w3'] == [(1, 2, 8)]
            accs = {n: id(g.layers_cache[n]['acc']) for n in layers}
            assert accs['gate'] not in (accs['e1.w1'], accs['e1.w3'], accs['e0.w1'], accs['e0.w3'])
            assert g.layers_cache['e1.w1']['nsamples'] == 1 and g.layers_cache['gate']['nsamples'] == 2
            # the expert that never ran still sits in the router's group: settling it gives it an (empty) Hessian of its own
            assert g._group_of['e2.w1'] == g._group_of['gate']
            g._settle_group(g._group_of['gate'])
            assert g._group_of['e2.w1'] != g._group_of['gate'] and g.layers_cache['e2.w1']['acc'].fed == []


def test_block_forward_counts_passes():
    from toy_model import ToyModel, calib_input
    from llmc_amd.compression.quantization.base_blockwise_quantization import BaseBlockwiseQuantization as B
    model = ToyModel(hidden=32, inner=48, n_blocks=1, dtype=torch.float32)
    b = B.__new__(B)
    b.input = calib_input(model, n_seq=3, seq=4)
    b.block_forward(model.get_blocks()[0])
    assert b._fwd_pass == 3


def test_true_sequential_first_pass_feeds_only_the_first_subset(monkeypatch):
    """under true_sequential the Hessians of subsets 2.. are re-accumulated after the earlier subsets are
    quantized (base_blockwise_quantization.py:506-526); the first pass must not compute them."""
    import llmc_amd.compression.quantization.gptq as gq

    class FakeAcc:
        def __init__(self, K, dev, defer=True, exact_diag=True):
            self.K, self.nsamples, self.fed = K, 0, []
            self.H = torch.zeros(K, K)

        def add(self, inp):
            self.fed.append(inp)
            self.nsamples += inp.shape[0]
    monkeypatch.setattr(gq, 'HessianAccumulator', FakeAcc)
    lin = lambda k: torch.nn.Linear(k, 4, bias=False)  # noqa: E731
    layers = {'q': lin(8), 'k': lin(8), 'o': lin(8), 'down': lin(16)}
    subsets = [{'layers': {'q': layers['q'], 'k': layers['k']}}, {'layers': {'o': layers['o']}}, {'layers': {'down': layers['down']}}]

    class M:
        def get_block_linears(self, b):
            return layers

        def get_subsets_in_block(self, b):
            return subsets
    for seq in (True, False):
        g = gq.GPTQ.__new__(gq.GPTQ)
        g.layers_cache, g._groups, g._group_of, g.model, g.true_sequential = {}, {}, {}, M(), seq
        g.block_init(None)
        h = torch.randn(1, 3, 8)
        for n in ('q', 'k', 'o'):
            g.add_batch(layers[n], n, h if n != 'o' else torch.randn(1, 3, 8), None)
        g.add_batch(layers['down'], 'down', torch.randn(1, 3, 16), None)
        fed = {n: len(g.layers_cache[n]['acc'].fed) for n in layers}
        if seq:
            assert fed == {'q': 1, 'k': 1, 'o': 0, 'down': 0}
            g.subset_init(subsets[1])                       # rehook_next_subset
            g.add_batch(layers['q'], 'q', h, None)          # a stray hook of an earlier subset is ignored
            g.add_batch(layers['o'], 'o', torch.randn(1, 3, 8), None)
            assert len(g.layers_cache['o']['acc'].fed) == 1
        else:
            assert fed == {'q': 1, 'k': 1, 'o': 1, 'down': 1}


def test_spqr_surface_and_config_errors_without_gpu():
    """SpQR keeps the reference's constructor / method surface (spqr.py:18-398) and refuses what the reference cannot run
    either (a symmetric weight quantizer) or what is not built (per_tensor second-level quantizers)."""
    sig = lambda f: list(inspect.signature(f).parameters)  # noqa: E731
    assert 'SpQR' in ALGO_REGISTRY.keys() and ALGO_REGISTRY['SpQR'] is Q.SpQR
    assert sig(Q.SpQR.__init__)[:6] == ['self', 'model', 'quant_config', 'input', 'padding_mask', 'config']
    assert sig(Q.SpQR.w_qdq) == ['self', 'module', 'wquantizer']
    q2 = dict(bit=3, symmetric=False, granularity='per_group', group_size=16, round_zp=False)
    special = dict(actorder=True, percdamp=1, blocksize=128, true_sequential=True, relative_threshold='inf',
                   simplified_outliers=False, scale=dict(q2), zero=dict(q2))
    s = Q.SpQR.__new__(Q.SpQR)
    s.quant_config = {'special': special}
    s.wquantizer = Q.IntegerQuantizer(4, False, 'per_group', group_size=16, round_zp=False)
    s.add_quant_config()
    import math
    assert s.relative_threshold == math.inf and s.need_perm and s.scfg.group_size == 16 and s.scfg.scale_bit == 3
    s.wquantizer = Q.IntegerQuantizer(4, True, 'per_group', group_size=16, round_zp=False)
    with pytest.raises(NotImplementedError):
        s.add_quant_config()
    s.wquantizer = Q.IntegerQuantizer(4, False, 'per_group', group_size=16, round_zp=False)
    s.quant_config = {'special': dict(special, scale=dict(q2, granularity='per_tensor'))}
    with pytest.raises(NotImplementedError):
        s.add_quant_config()
    with pytest.raises(AssertionError):
        s.deploy('real_quant')


def test_blocked_output_index_map_without_gpu():
    """awq_ops.unblock_y: the tile-blocked image (tile, wave, accumulator, register, lane) back to [N, R]."""
    from llmc_amd.compression.quantization.awq_ops import unblock_y
    N, R, ntm, ntn = 300, 520, 2, 3
    ref = torch.arange(ntm * 256 * ntn * 256, dtype=torch.int32).reshape(ntm * 256, ntn * 256)
    idx = torch.arange(ntm * ntn * 65536)
    j, lane, piece, wv, t = idx % 8, (idx // 8) % 64, (idx // 512) % 32, (idx // 16384) % 4, idx // 65536
    v, n, m, wm, wn, tm, tn = piece % 2, (piece // 2) % 4, piece // 8, wv // 2, wv % 2, t // ntn, t % ntn
    r = 8 * v + j
    tok = tm * 256 + wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    col = tn * 256 + wn * 128 + n * 32 + (lane & 31)
    buf = ref[tok, col]
    assert torch.equal(unblock_y(buf, N, R), ref[:N, :R])


def test_awq_and_autoclipper_route_by_quantizer_kind():
    """Host-side routing (no GPU): the fused kernels take W4A16-style integer row / group ranges only; quantized activations, FP8
    and per_tensor weight quantizers, mse ranges and clip_version v2 go to the general routes (awq.py:_fusable_wquantizer,
    auto_clip.py:_fused_route); clip v2 with per_group weights is refused where the configuration is read."""
    import pytest
    import torch
    from llmc_amd.compression.quantization import FloatQuantizer, IntegerQuantizer
    from llmc_amd.compression.quantization.auto_clip import AutoClipper
    from llmc_amd.compression.quantization.awq import Awq

    def awq(wq, w_only=True):
        a = Awq.__new__(Awq)
        a.wquantizer, a.w_only, a.padding_mask, a.awq_bs = wq, w_only, None, None
        return a
    lin = torch.nn.Linear(64, 64, bias=False)
    x = [torch.zeros(2, 4, 64)]
    g128 = IntegerQuantizer(4, True, 'per_group', group_size=128)
    assert awq(g128)._fused_route_ok({'o': lin}, x, lin, {})
    assert awq(IntegerQuantizer(8, True, 'per_channel'))._fused_route_ok({'o': lin}, x, lin, {})
    for wq in (IntegerQuantizer(8, True, 'per_tensor'), IntegerQuantizer(4, True, 'per_group', group_size=128, calib_algo='mse'),
               FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True), FloatQuantizer('e5m2', True, 'per_channel', use_qtorch=True)):
        assert not awq(wq)._fused_route_ok({'o': lin}, x, lin, {}), repr(wq)
    assert not awq(g128, w_only=False)._fused_route_ok({'o': lin}, x, lin, {})

    def clipper(wq, w_only=True, ver='v1', aq=None, **kw):
        return AutoClipper(w_only=w_only, wquantizer=wq, aquantizer=aq, clip_version=ver, clip_sym=True, save_clip=False,
                           padding_mask=None, **kw)
    assert clipper(g128)._fused_route()
    a8 = IntegerQuantizer(8, True, 'per_token')
    for c in (clipper(IntegerQuantizer(4, True, 'per_channel')), clipper(IntegerQuantizer(4, True, 'per_group', group_size=256)),
              clipper(g128, w_only=False, aq=a8), clipper(FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True), w_only=False, aq=a8),
              clipper(IntegerQuantizer(4, True, 'per_channel', calib_algo='learnable'), ver='v2')):
        assert not c._fused_route()
    with pytest.raises(NotImplementedError, match='external_ranges'):
        clipper(IntegerQuantizer(4, True, 'per_group', group_size=128, calib_algo='learnable'), ver='v2')
    clipper(IntegerQuantizer(4, True, 'per_group', group_size=128, calib_algo='learnable'), ver='v2', external_ranges=True)
