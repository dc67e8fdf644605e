"""AutoClipper clip_version v2's durable half (reference auto_clip.py:213-256, quant.py:205-224) against outputs of the
reference itself (tests/golden/clip_v2.npz, oracle/make_golden.py suite_clip_v2): the logit range factors apply_clip stores
and the weights a `calib_algo: learnable` quantizer fake-quantizes with them. The CPU half checks the factor arithmetic (pure
torch host logic); the GPU half runs apply_clip + w_qdq through the HIP quantizer."""
import os

import numpy as np
import pytest
import torch

from conftest import report

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'clip_v2.npz'))
DT = {'bf16': torch.bfloat16, 'f16': torch.float16}
NAMES = [str(n) for n in G['names']]


def case(name, device):
    p = name + '/'
    sym, gs, clip_sym = (int(v) for v in G[p + 'meta'])
    dt = DT[str(G[p + 'dt'])]
    t = lambda k: torch.from_numpy(G[p + k]).to(dt).to(device)
    from llmc_amd.compression.quantization.quant import IntegerQuantizer
    from llmc_amd.compression.quantization.auto_clip import AutoClipper
    kw = dict(group_size=gs) if gs else {}
    wq = IntegerQuantizer(4, bool(sym), 'per_group' if gs else 'per_channel', calib_algo='learnable', **kw)
    ac = AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v2', clip_sym=bool(clip_sym),
                     save_clip=True, padding_mask=None, external_ranges=True)
    w = t('w')
    layer = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device=device, dtype=dt)
    layer.weight.data = w.clone()
    return ac, wq, layer, t, bool(clip_sym)


@pytest.mark.parametrize('name', NAMES)
def test_clip_factors_match_reference_cpu(name):
    ac, wq, layer, t, clip_sym = case(name, 'cpu')
    up, low = ac.get_clip_factor(0, layer, t('min'), t('max'), 'fc')
    assert torch.equal(up.float(), t('up_factor').float())       # same torch ops on the same CPU: bit-exact
    if clip_sym:
        assert low is None
    else:
        assert torch.equal(low.float(), t('low_factor').float())


def test_v2_search_refuses_per_group_like_the_reference():
    """The v2 range search runs for per_channel / per_tensor weights (GPU: test_clip_wide_gpu.py). With per_group weights the
    reference raises inside its own quantizer (quant.py:701); here the configuration is refused where it is read, and the
    search itself refuses when a caller with external ranges asks for it anyway."""
    from llmc_amd.compression.quantization.auto_clip import AutoClipper
    from llmc_amd.compression.quantization.quant import IntegerQuantizer
    pg = [n for n in NAMES if '_g' in n][0]
    ac, wq, layer, t, _ = case(pg, 'cpu')
    with pytest.raises(NotImplementedError, match='per_group'):
        ac.auto_clip_layer(0, 'fc', layer.weight, [torch.zeros(1, 4, layer.weight.shape[1], dtype=layer.weight.dtype)], 20, 0.5, 4)
    with pytest.raises(NotImplementedError, match='external_ranges'):       # v2 + per_group without ranges of the caller's own
        AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v2', clip_sym=True, save_clip=False,
                    padding_mask=None)
    pc = IntegerQuantizer(4, True, 'per_channel', calib_algo='learnable')
    AutoClipper(w_only=True, wquantizer=pc, aquantizer=None, clip_version='v2', clip_sym=True, save_clip=False,
                padding_mask=None)                                           # per_channel: the search is available
    with pytest.raises(Exception, match='clip version'):
        AutoClipper(w_only=True, wquantizer=wq, aquantizer=None, clip_version='v3', clip_sym=True, save_clip=False,
                    padding_mask=None)


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_apply_clip_v2_and_learnable_w_qdq_gpu(name):
    ac, wq, layer, t, clip_sym = case(name, 'cuda')
    ac.apply_clip(0, layer, t('min'), t('max'), 'fc')
    up, low = layer.buf_upbound_factor, layer.buf_lowbound_factor
    gu = t('up_factor')
    # the factors go through a 16-bit divide and an fp32 log evaluated by the GPU's libm: allow the last 16-bit place
    fin = torch.isfinite(gu.float())
    ulp = 2.0 ** (-7 if layer.weight.dtype == torch.bfloat16 else -10)
    du = ((up.float() - gu.float()).abs()[fin] / gu.float().abs().clamp(min=1.0)[fin]).max().item()
    assert torch.equal(torch.isfinite(up.float()), fin) and du <= 2 * ulp, du
    if not clip_sym:
        gl = t('low_factor')
        finl = torch.isfinite(gl.float())
        dl = ((low.float() - gl.float()).abs()[finl] / gl.float().abs().clamp(min=1.0)[finl]).max().item()
        assert dl <= 2 * ulp, dl
    else:
        assert low is None
    assert ac.weight_clips[0]['fc.weight_quantizer.upbound_factor'].shape == up.shape

    # w_qdq with the reference's own factors: isolates the quantizer from the libm difference above
    args = {'upbound_factor': gu, 'lowbound_factor': None if clip_sym else t('low_factor')}
    fq = wq.fake_quant_weight_dynamic(layer.weight.data, args)
    ref = t('w_qdq')
    agree = (fq.float() == ref.float()).float().mean().item()
    report(f'clip_v2/{name}', w_qdq_agree=agree, up_factor_rel=du)
    assert agree >= 0.999, agree
    # and with no factors a learnable quantizer is the plain min/max one
    fq0 = wq.fake_quant_weight_dynamic(layer.weight.data, {'upbound_factor': None, 'lowbound_factor': None})
    assert torch.equal(fq0.float(), t('w_qdq_nofactor').float())
