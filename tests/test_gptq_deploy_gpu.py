"""GPTQ.w_qdq / GPTQ.w_q (gptq.py:412-452) against the reference's own outputs in tests/golden/gptq.npz:
`final_w` (layer.weight after update_layer_with_transformed_weights), `buf_scales` / `buf_zeros` with the dtypes the
reference leaves them in (fp32 after dynamic-group GPTQ, model dtype RTN qparams with static groups,
0-dim zeros for symmetric), `w_qdq`, and `w_q_codes / w_q_scales / w_q_zeros` where the layer is exportable."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TD = {'f16': torch.float16, 'bf16': torch.bfloat16, 'torch.float16': torch.float16, 'torch.bfloat16': torch.bfloat16,
      'torch.float32': torch.float32}


def build_layer(g, p):
    bit, sym, gs, actorder, static_groups, R, K, qmin, qmax = g[p + 'meta']
    sym, gs, actorder, static_groups, R, K = bool(sym), int(gs), bool(actorder), bool(static_groups), int(R), int(K)
    layer = torch.nn.Linear(K, R, bias=False)
    layer.weight.data = torch.from_numpy(g[p + 'final_w']).float()          # fp32 after GPTQ (G3)
    sdt = TD[str(g[p + 'buf_scales_dtype'])]
    layer.register_buffer('buf_scales', torch.from_numpy(g[p + 'buf_scales']).to(sdt).reshape(-1, 1))
    bz = g[p + 'buf_zeros']
    if bz.size:
        layer.register_buffer('buf_zeros', torch.from_numpy(bz).to(sdt).reshape(-1, 1))
    else:
        layer.register_buffer('buf_zeros', torch.tensor(0.0))
    layer.register_buffer('buf_qmax', torch.tensor(qmax))
    layer.register_buffer('buf_qmin', torch.tensor(qmin))
    perm = g[p + 'perm']
    if perm.size:
        layer.register_buffer('buf_perm', torch.from_numpy(perm))
        layer.register_buffer('buf_invperm', torch.argsort(torch.from_numpy(perm)))
    return layer.cuda(), dict(bit=int(bit), sym=sym, gs=gs, actorder=actorder, static_groups=static_groups)


def test_w_qdq_and_w_q_match_reference_goldens():
    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.gptq import GPTQ
    g = load_golden('gptq+more')
    checked_wq = 0
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        layer, c = build_layer(g, p)
        gran = str(g[p + 'gran'])
        kw = {'group_size': c['gs']} if gran == 'per_group' else {}
        wq = IntegerQuantizer(c['bit'], c['sym'], gran, **kw)
        model_dtype = TD[str(g[p + 'dt'])]
        need_perm = gran == 'per_group' and not c['static_groups'] and c['actorder']
        this = types.SimpleNamespace(need_perm=need_perm, model_dtype=model_dtype)
        fq = GPTQ.w_qdq(this, layer, wq)
        assert fq.dtype == TD[str(g[p + 'w_qdq_dtype'])], name
        np.testing.assert_array_equal(fq.float().cpu().numpy().view(np.uint32), g[p + 'w_qdq'].view(np.uint32),
                                      err_msg=name)
        if need_perm:
            assert (p + 'w_q_codes') not in g.files       # the reference asserts out of deploy here (gptq.py:455-457)
            continue
        codes, scales, zeros = GPTQ.w_q(this, layer, wq)
        ref_codes = g[p + 'w_q_codes']
        assert codes.dtype == (torch.int32 if c['bit'] != 8 else (torch.int8 if c['sym'] else torch.uint8)), name
        np.testing.assert_array_equal(codes.cpu().numpy().astype(np.int32), ref_codes, err_msg=name)
        assert scales.dtype == model_dtype, name                       # scales.to(model_dtype) first (G2)
        np.testing.assert_array_equal(scales.float().cpu().numpy().view(np.uint32).reshape(-1),
                                      g[p + 'w_q_scales'].view(np.uint32).reshape(-1), err_msg=name)
        rz = g[p + 'w_q_zeros']
        if rz.size:
            assert zeros.dtype == codes.dtype, name
            np.testing.assert_array_equal(zeros.cpu().numpy().astype(np.int32).reshape(-1), rz.reshape(-1),
                                          err_msg=name)
        else:
            assert zeros is None, name
        checked_wq += 1
    assert checked_wq >= 3
