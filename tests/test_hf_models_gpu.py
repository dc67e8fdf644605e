"""The host classes on REAL Hugging Face blocks (random init): the plumbing of BASELINE.json configs[0] (RTN W8A16
per-channel on the OPT-125M architecture, with biases) and the Llama subset table (GQA shapes, attention kwargs, rotary
embeddings) through GPTQ and AWQ — tests/hf_adapters.py mirrors llmc/models/opt.py:53-90 and llama.py:52-91 and
captures the first block's inputs like llmc/models/base_model.py:171-189."""
import copy

import numpy as np
import pytest
import torch

from oracle import quant_ref as Qr

pytestmark = pytest.mark.gpu


class Cfg(dict):
    __getattr__ = dict.get


def test_rtn_w8a16_per_channel_on_the_opt_125m_architecture_bit_exact():
    import hf_adapters as H
    import llmc_amd.compression.quantization as Q
    model = H.opt_125m_shaped(torch.float16)
    inp = model.collect_first_block_input(H.calib_ids(2, 64, 512))
    w0 = {n: m.weight.data.clone() for n, m in model.get_block_linears(model.get_blocks()[11]).items()}
    b0 = model.get_blocks()[11].fc2.bias.data.clone()
    qc = Cfg(weight=Cfg(bit=8, symmetric=True, granularity='per_channel'))      # ci_check-style W8A16 (configs[0])
    config = Cfg(calib=Cfg(seq_len=64), model=Cfg(type='Opt'))
    algo = Q.RTN(model, qc, inp, None, config)
    algo.run_block_loop()
    algo.deploy('fake_quant')
    blk = model.get_blocks()[11]
    for n in ('self_attn.q_proj', 'fc1', 'fc2'):
        m = blk.get_submodule(n)
        assert type(m).__name__ == 'EffcientFakeQuantLinear' and m.weight.dtype == torch.float16
        w = w0[n].float().numpy()
        ref, _, _ = Qr.fake_quant_dynamic(w, 'f16', True, -128.0, 127.0)      # per_channel: one group per row
        np.testing.assert_array_equal(m.weight.data.float().cpu().numpy(), ref, err_msg=n)
    assert torch.equal(blk.fc2.bias.data.cpu(), b0)                               # biases ride along untouched
    # the deployed model still runs end to end on the GPU (the wrappers' forward is the HIP GEMM + bias epilogue)
    model.model.cuda()
    ids = H.calib_ids(1, 64, 512, seed=5)[0].cuda()
    with torch.no_grad():
        logits = model.model(ids).logits
    assert torch.isfinite(logits).all()


def _llama_with_inputs(n=4, seq=128):
    import hf_adapters as H
    model = H.tiny_llama(torch.bfloat16)
    inp = model.collect_first_block_input(H.calib_ids(n, seq, 160))
    return model, inp


def test_gptq_on_llama_blocks_with_gqa_and_attention_kwargs():
    import llmc_amd.compression.quantization as Q
    model, inp = _llama_with_inputs()
    ref_blocks = copy.deepcopy(model.get_blocks())
    qc = Cfg(weight=Cfg(bit=4, symmetric=False, granularity='per_group', group_size=128),
             special=Cfg(actorder=True, static_groups=False, percdamp=0.01, blocksize=128, true_sequential=True),
             quant_out=True)
    config = Cfg(calib=Cfg(seq_len=128), model=Cfg(type='Llama'))
    algo = Q.GPTQ(model, qc, copy.deepcopy(inp), None, config)
    assert algo.has_gqa and algo.num_key_value_groups == 2
    algo.run_block_loop()
    blk = model.get_blocks()[0]
    shapes = {'self_attn.q_proj': (256, 256), 'self_attn.k_proj': (128, 256), 'self_attn.v_proj': (128, 256),
              'self_attn.o_proj': (256, 256), 'mlp.gate_proj': (512, 256), 'mlp.up_proj': (512, 256), 'mlp.down_proj': (256, 512)}
    for n, shp in shapes.items():
        m = blk.get_submodule(n)
        assert tuple(m.weight.shape) == shp and m.weight.dtype == torch.float32 and torch.isfinite(m.weight).all(), n
        assert m.buf_scales.shape == (shp[0] * shp[1] // 128, 1) and m.buf_scales.dtype == torch.float32
        assert sorted(m.buf_perm.tolist()) == list(range(shp[1]))
    # q / k / v saw the very same tensor: one shared permutation (one Hessian); o_proj has its own
    assert torch.equal(blk.self_attn.q_proj.buf_perm, blk.self_attn.k_proj.buf_perm)
    assert torch.equal(blk.mlp.gate_proj.buf_perm, blk.mlp.up_proj.buf_perm)
    # GPTQ beats round-to-nearest on the layer's own calibration inputs (what it minimises)
    x = inp['data'][0].cuda()
    rb = ref_blocks[0].cuda()
    h = rb.input_layernorm(x).reshape(-1, 256).float()
    w = rb.self_attn.q_proj.weight.data.float()
    algo.deploy('fake_quant')
    wq = model.get_blocks()[0].cuda().self_attn.q_proj.weight.data.float()
    rtn = Q.IntegerQuantizer(4, False, 'per_group', group_size=128).fake_quant_weight_dynamic(rb.self_attn.q_proj.weight.data).float()
    e_gptq = (h @ (wq - w).T).norm()
    e_rtn = (h @ (rtn - w).T).norm()
    assert e_gptq < e_rtn, (float(e_gptq), float(e_rtn))


def test_awq_on_llama_blocks_inspecting_the_attention_module():
    """q/k/v are searched through `inspect = block.self_attn` with the block's kwargs (llama.py:62): the general route,
    its Linear calls on the k-tiled HIP GEMM with the shared activation packed once per grid point."""
    import llmc_amd.compression.quantization as Q
    from llmc_amd.compression.quantization import awq_ops
    import hf_adapters as H
    model = H.tiny_llama(torch.bfloat16)
    g = torch.Generator().manual_seed(4)
    for blk in model.get_blocks():      # outlier channels (random-init norms are all ones: the search would return s = 1)
        for ln in (blk.input_layernorm, blk.post_attention_layernorm):
            ln.weight.data *= torch.exp(0.7 * torch.randn(256, generator=g)).to(torch.bfloat16)
            ln.weight.data[torch.randperm(256, generator=g)[:4]] *= 30
    inp = model.collect_first_block_input(H.calib_ids(4, 128, 160))
    ref_blocks = copy.deepcopy(model.get_blocks())
    inp1 = {'data': [torch.cat(inp['data'], dim=0)], 'kwargs': [inp['kwargs'][0]]}     # calib.bs = -1: one batch
    qc = Cfg(weight=Cfg(bit=4, symmetric=True, granularity='per_group', group_size=128),
             special=Cfg(trans=True, trans_version='v2', weight_clip=True, clip_sym=True))
    config = Cfg(calib=Cfg(seq_len=128), model=Cfg(type='Llama'))
    calls = {'kt': 0}
    orig = awq_ops.ktile_pack

    def counting(m):
        calls['kt'] += 1
        return orig(m)
    awq_ops.ktile_pack = counting
    try:
        algo = Q.Awq(model, qc, inp1, None, config)
        algo.run_block_loop()
    finally:
        awq_ops.ktile_pack = orig
    assert calls['kt'] > 0                                   # the fast GEMM ran under the inspected attention / MLP modules
    # scale folding keeps the float function of the block (before quantisation), up to clipping
    x = inp1['data'][0].cuda()
    kw = {k: (v.cuda() if torch.is_tensor(v) else tuple(t.cuda() for t in v) if isinstance(v, tuple) else v)
          for k, v in inp1['kwargs'][0].items()}
    b_new, b_old = model.get_blocks()[0].cuda(), ref_blocks[0].cuda()
    with torch.no_grad():
        y_new, y_old = b_new(x, **kw), b_old(x, **kw)
    y_new = (y_new[0] if isinstance(y_new, tuple) else y_new).float()
    y_old = (y_old[0] if isinstance(y_old, tuple) else y_old).float()
    assert ((y_new - y_old).norm() / y_old.norm()).item() < 0.1
    s = ref_blocks[0].input_layernorm.weight.data.float().cpu() / model.get_blocks()[0].input_layernorm.weight.data.float().cpu()
    assert s.min() > 0 and s.max() / s.min() > 1.05          # a non-trivial scale was folded into the layer norm
