"""Drive the REFERENCE'S OWN pipeline — llmc/__main__.py:28-176 `main(config)`: MODEL_REGISTRY adapter (models/llama.py:52-91,
models/opt.py:53-90), BaseDataset + wikitext2_gptq preproc, collect_first_block_input, ALGO_REGISTRY[method](...).run_block_loop()
(compression/blockwise_optimization.py:53-61), PerplexityEval — on a random-init model saved with `save_pretrained`, twice:

  --arm ref    the reference untouched (oracle/_ref on a CPU-only host, oracle/_ref_gpu = plain copy on a GPU box)
  --arm ours   the same process, the same reference code, with ONE extra line before main():
                   llmc_amd.register_into(ALGO_REGISTRY)            

and dump what every Linear layer ends with (weight, buf_scales, buf_zeros, buf_perm) plus the perplexity the reference's own
evaluator printed. tests/test_ref_pipeline_gpu.py compares the two dumps. Test infrastructure: never imported by the product.

Nothing of the reference's control plane is rebuilt here: this file only fabricates the assets the reference expects on disk
(a checkpoint directory, a tokenizer, a `datasets` directory) and stubs third-party imports the image lacks (librosa,
torchvision, human_eval, lmms_eval, diffusers — used by adapters / evaluators this run never touches)."""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORDS = 500


def make_assets(out, arch, seed=0, n_layers=2):
    """checkpoint + tokenizer + dataset under `out` (idempotent). arch: 'llama' (GQA, SwiGLU, RMSNorm) or 'opt'
    (OPT-125M widths: hidden 768, ffn 3072, 12 heads, biases, LayerNorm; `n_layers` blocks instead of 12)."""
    import torch
    from datasets import Dataset
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    mdir, ddir = os.path.join(out, 'model'), os.path.join(out, 'data')
    if os.path.exists(os.path.join(out, '.done')):
        return mdir, ddir
    os.makedirs(out, exist_ok=True)
    vocab = {'<unk>': 0, '<s>': 1, '</s>': 2, '<pad>': 3}
    for i in range(WORDS):
        vocab[f'w{i}'] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token='<unk>', bos_token='<s>', eos_token='</s>', pad_token='<pad>')
    torch.manual_seed(seed)
    if arch == 'llama':
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(vocab_size=len(vocab), hidden_size=256, intermediate_size=512, num_hidden_layers=n_layers,
                          num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512,
                          bos_token_id=1, eos_token_id=2, pad_token_id=3, tie_word_embeddings=False, initializer_range=0.08)
        model = LlamaForCausalLM(cfg)
    else:
        from transformers import OPTConfig, OPTForCausalLM
        cfg = OPTConfig(vocab_size=len(vocab), hidden_size=768, ffn_dim=3072, num_hidden_layers=n_layers, num_attention_heads=12,
                        max_position_embeddings=512, word_embed_proj_dim=768, bos_token_id=1, eos_token_id=2, pad_token_id=3, init_std=0.05)
        model = OPTForCausalLM(cfg)
    # channel scales with a spread plus a few outlier channels (SURVEY 8d's activation recipe, applied at the embedding: the
    # norms in front of q|k|v and gate|up keep relative channel scales), so that actorder has a well-determined order and
    # the AWQ search an interior optimum instead of ties decided by rounding noise
    with torch.no_grad():
        emb = model.get_input_embeddings().weight
        emb.mul_(torch.exp(0.5 * torch.randn(emb.shape[1])))
        emb[:, torch.randperm(emb.shape[1])[:4]] *= 8.0
    model = model.to(torch.float16)
    model.save_pretrained(mdir)
    fast.save_pretrained(mdir)
    g = torch.Generator().manual_seed(seed + 1)
    lines = []
    for _ in range(1200):
        n = int(torch.randint(5, 40, (1,), generator=g))
        ids = torch.randint(0, WORDS, (n,), generator=g)
        # a skewed unigram distribution: low ids are frequent
        ids = (ids.float() ** 2 / WORDS).long()
        lines.append(' '.join(f'w{int(i)}' for i in ids))
    Dataset.from_dict({'text': lines}).save_to_disk(ddir)
    open(os.path.join(out, '.done'), 'w').write('ok')
    return mdir, ddir


CONFIGS = {
    # ci_check/gptq_w_only.yml (the reference's own CI configuration), paths and model type filled in
    'gptq': dict(
        quant=dict(method='GPTQ', weight=dict(bit=4, symmetric=False, granularity='per_group', group_size=128),
                   special=dict(actorder=True, static_groups=False, percdamp=0.01, blocksize=128, true_sequential=True),
                   quant_out=True),
        calib=dict(name='wikitext2', download=False, n_samples=128, bs=1, seq_len=64, preproc='wikitext2_gptq')),
    # ci_check/awq_w4a16_fakequant_eval.yml
    'awq': dict(
        quant=dict(method='Awq', weight=dict(bit=4, symmetric=False, granularity='per_group', group_size=128),
                   special=dict(trans=True, trans_version='v2', weight_clip=True, clip_sym=False)),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq')),
    # configs[0] of BASELINE.json: RTN W8A16 per-channel (plumbing)
    'rtn': dict(
        quant=dict(method='RTN', weight=dict(bit=8, symmetric=True, granularity='per_channel')),
        calib=None),
    # mixed precision (base_blockwise_quantization.py:137-144, 910-932): two layers of block 0 and one full name stay float
    'rtn_mixed': dict(
        quant=dict(method='RTN', weight=dict(bit=4, symmetric=True, granularity='per_group', group_size=128)),
        calib=None,
        ignored_layers=dict(block_ids=[0], layer_names=['self_attn.q_proj', 'self_attn.v_proj'], speical_names=['@PREFIX@.1.self_attn.k_proj'])),
    # the export step of main() (llmc/__main__.py:95-144): deploy('vllm_quant') + save_model + update_vllm_quant_config for the
    # compressed-tensors layout (configs/quantization/backend/vllm/rtn_w4a16.yml), deploy('autoawq_quant') + save_model +
    # update_autoawq_quant_config for AutoAWQ's GEMM layout (backend/autoawq/awq_w4a16.yml); the saved checkpoints are compared
    'rtn_vllm': dict(
        quant=dict(method='RTN', weight=dict(bit=4, symmetric=True, granularity='per_group', group_size=128, need_pack=True)),
        calib=None, save=dict(save_vllm=True)),
    # configs/quantization/backend/vllm/gptq_w4a16.yml: the vLLM-exportable GPTQ variant of SURVEY 8d (sym, static groups, packed)
    'gptq_vllm': dict(
        quant=dict(method='GPTQ', weight=dict(bit=4, symmetric=True, granularity='per_group', group_size=128, need_pack=True),
                   special=dict(actorder=True, static_groups=True, percdamp=0.01, blocksize=128, true_sequential=True), quant_out=True),
        calib=dict(name='wikitext2', download=False, n_samples=128, bs=1, seq_len=64, preproc='wikitext2_gptq'), save=dict(save_vllm=True)),
    'awq_autoawq': dict(
        quant=dict(method='Awq', weight=dict(bit=4, symmetric=False, granularity='per_group', group_size=128, pack_version='gemm_pack'),
                   special=dict(trans=True, trans_version='v2', weight_clip=True, clip_sym=False)),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq'), save=dict(save_autoawq=True)),
    # configs/quantization/backend/vllm/fp8/rtn_fp8.yml as shipped (per_channel weights, per_token dynamic activations), and the
    # per-tensor form BASELINE configs[4] names (per_tensor weights, static per_tensor activations) — both exported for vLLM
    'rtn_fp8': dict(
        quant=dict(method='RTN', weight=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_channel', use_qtorch=True),
                   act=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_token', use_qtorch=True)),
        calib=None),      # no export: update_vllm_quant_config (export_vllm.py:33-42) asks every dynamic FP8 W-A config for weight.block_size
    'rtn_fp8_tensor': dict(
        quant=dict(method='RTN', weight=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_tensor', use_qtorch=True),
                   act=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_tensor', use_qtorch=True, static=True,
                            calib_algo='static_minmax')),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq'), save=dict(save_vllm=True)),
    # configs/quantization/methods/SpQR/spqr_w_only.yml: W4 g16 with 3-bit second-level statistics, outliers kept in fp
    'spqr': dict(
        quant=dict(method='SpQR', weight=dict(bit=4, symmetric=False, granularity='per_group', group_size=16, round_zp=False),
                   special=dict(actorder=True, percdamp=1, blocksize=128, true_sequential=True, relative_threshold=0.2,
                                simplified_outliers=False,
                                scale=dict(bit=3, symmetric=False, granularity='per_group', group_size=16, round_zp=False),
                                zero=dict(bit=3, symmetric=False, granularity='per_group', group_size=16, round_zp=False)),
                   quant_out=True),
        calib=dict(name='wikitext2', download=False, n_samples=128, bs=1, seq_len=64, preproc='wikitext2_gptq')),
    # AWQ with ACTIVATION quantization (awq.py:166-177, 223-224; auto_clip.py:276-281): configs/quantization/methods/Awq/
    # awq_w_a.yml's shape — W8 per_channel + A8 per_token dynamic, scale search and weight clip with quantized inputs
    'awq_w8a8': dict(
        quant=dict(method='Awq', weight=dict(bit=8, symmetric=True, granularity='per_channel'),
                   act=dict(bit=8, symmetric=True, granularity='per_token'),
                   special=dict(trans=True, trans_version='v2', weight_clip=True), quant_out=True),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq')),
    # more shipped files of the path's families, as they are: methods/GPTQ/gptq_owq_w_only.yml (OWQ: the most sensitive input
    # channels of every layer stay in floating point), methods/RTN/rtn_w_a_pertensor_static.yml (W8A8 with static_hist activation
    # ranges), methods/Awq/awq_w_a_mix_bits.yml (W4A4 with down_proj at W8A8 through mix_bits, symmetric clipping),
    # methods/RTN/rtn_w_a_block.yml (FP8 128 x 128 block-wise weights, FP8 activations in groups of 128)
    'gptq_owq': dict(
        quant=dict(method='GPTQ', weight=dict(bit=4, symmetric=False, granularity='per_group', group_size=128),
                   special=dict(actorder=False, static_groups=False, percdamp=0.01, blocksize=128, true_sequential=True, owq=True,
                                n_outs=[6, 6, 6, 6, 2, 2, 6]), quant_out=True),
        calib=dict(name='wikitext2', download=False, n_samples=128, bs=1, seq_len=64, preproc='wikitext2_gptq')),
    'rtn_static_hist': dict(
        quant=dict(method='RTN', weight=dict(bit=8, symmetric=True, granularity='per_channel', group_size=-1),
                   act=dict(bit=8, symmetric=True, granularity='per_tensor', static=True, calib_algo='static_hist')),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq')),
    'awq_mix_w_a': dict(
        quant=dict(method='Awq', weight=dict(bit=4, symmetric=False, granularity='per_channel'),
                   act=dict(bit=4, symmetric=False, granularity='per_token'),
                   mix_bits=dict(setting_0=dict(layer_name=['down_proj'], do_quant=True,
                                                weight=dict(bit=8, symmetric=False, granularity='per_channel'),
                                                act=dict(bit=8, symmetric=False, granularity='per_token'))),
                   special=dict(trans=True, trans_version='v2', weight_clip=True, clip_sym=True)),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq')),
    'rtn_fp8_block': dict(
        quant=dict(method='RTN', weight=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_block', block_size=128,
                                              use_qtorch=True),
                   act=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_group', group_size=128, use_qtorch=True)),
        calib=None),
    # configs/quantization/combination/awq_comb_omni/w8a8/step_1_awq.yml: the configuration that selects AutoClipper clip_version v2
    # (learnable-range weights, asymmetric per_channel W8 + per_token A8, scales and clip factors saved for OmniQuant's second step)
    'awq_v2_w8a8': dict(
        quant=dict(method='Awq', weight=dict(bit=8, symmetric=False, granularity='per_channel', group_size=-1, calib_algo='learnable'),
                   act=dict(bit=8, symmetric=False, granularity='per_token', calib_algo='minmax'),
                   special=dict(trans=True, trans_version='v2', weight_clip=True, clip_version='v2', save_scale=True,
                                scale_path='@ASSETS@/v2_scale_@ARM@', save_clip=True, clip_path='@ASSETS@/v2_clip_@ARM@')),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq')),
    # configs/quantization/backend/vllm/fp8/awq_fp8_static.yml (the parent of BASELINE configs[4]): FP8 e4m3 per_tensor
    # weights, FP8 e4m3 per_tensor STATIC activations, trans v2 + weight clip. float_quantize of the reference arm is bound
    # to the restated qtorch (oracle/quant_ref.py) — qtorch itself is not installable here
    'awq_fp8': dict(
        quant=dict(method='Awq', weight=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_tensor', use_qtorch=True),
                   act=dict(quant_type='float-quant', bit='e4m3', symmetric=True, granularity='per_tensor', use_qtorch=True, static=True,
                            calib_algo='static_minmax'),   # the shipped yml leaves the default 'minmax', which quant.py:573-574 refuses
                   special=dict(trans=True, trans_version='v2', weight_clip=True), quant_out=True),
        calib=dict(name='wikitext2', download=False, n_samples=32, bs=-1, seq_len=64, preproc='wikitext2_gptq')),
}


def bind_restated_qtorch():
    """llmc.compression.quantization.quant.float_quantize := the restatement of QPyTorch's float_quantize (test
    infrastructure: the reference arm only; tensors take the trip through the host)."""
    import numpy as np
    import torch
    import llmc.compression.quantization.quant as qmod
    sys.path.insert(0, ROOT)
    from oracle import quant_ref as QR

    def float_quantize(x, e, m, rounding='nearest'):
        assert rounding == 'nearest'
        y = QR.qtorch_float_quantize(x.detach().float().cpu().numpy(), e, m)
        return torch.from_numpy(np.ascontiguousarray(y)).reshape(x.shape).to(x.device)
    qmod.float_quantize = float_quantize


def build_config(method, arch, mdir, ddir, save_path, assets='', arm=''):
    c = json.loads(json.dumps(CONFIGS[method]).replace('@ASSETS@', assets).replace('@ARM@', arm))
    cfg = {'base': {'seed': 0},
           'model': {'type': 'Llama' if arch == 'llama' else 'Opt', 'path': mdir, 'torch_dtype': 'auto'},
           'eval': {'eval_pos': ['fake_quant'], 'name': 'wikitext2', 'download': False, 'path': ddir, 'bs': 1, 'seq_len': 64,
                    'inference_per_block': False},
           'quant': c['quant'],
           'save': {'save_fake': False, 'save_path': save_path}}
    if c['calib']:
        cfg['calib'] = dict(c['calib'], path=ddir, seed=0)
    if c.get('save'):
        cfg['save'].update(c['save'])
    if c.get('ignored_layers'):
        il = c['ignored_layers']
        prefix = 'model.layers' if arch == 'llama' else 'model.decoder.layers'
        il['speical_names'] = [n.replace('@PREFIX@', prefix) for n in il['speical_names']]
        cfg['ignored_layers'] = il
    return cfg


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        v = type(n, (), {'__init__': lambda self, *a, **k: None, '__call__': lambda self, *a, **k: None})
        setattr(self, n, v)
        return v


ALLOWED_STUBS = ('librosa', 'torchvision', 'human_eval', 'lmms_eval', 'diffusers', 'qtorch', 'fast_hadamard_transform',
                 'flash_attn', 'timm', 'decord', 'av', 'llava', 'vllm', 'lightllm', 'sglang', 'qwen_vl_utils')


def import_reference_main(ref_dir):
    """`import llmc.__main__` from `ref_dir`, stubbing third-party modules the image lacks — after transformers has
    decided for itself which optional back ends exist (a stub that transformers can see would be mistaken for the real one)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_shims'))
    sys.path.insert(0, ref_dir)
    import transformers  # noqa: F401
    from transformers.utils import import_utils as IU
    for n in dir(IU):
        if n.startswith('is_') and n.endswith('_available'):
            try:
                getattr(IU, n)()
            except Exception:       # noqa: BLE001
                pass
    import transformers.models.llama.modeling_llama  # noqa: F401
    import transformers.models.opt.modeling_opt  # noqa: F401
    stubbed = []
    for _ in range(80):
        try:
            import llmc.__main__ as M
            return M, stubbed
        except ModuleNotFoundError as e:
            name = e.name or ''
            if name.split('.')[0] not in ALLOWED_STUBS:
                raise
            parts = name.split('.')
            for i in range(1, len(parts) + 1):
                nm = '.'.join(parts[:i])
                if nm not in sys.modules:
                    m = _Stub(nm)
                    m.__path__ = []
                    sys.modules[nm] = m
                    if i > 1:
                        setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)
            stubbed.append(name)
            for k in [k for k in sys.modules if k == 'llmc' or k.startswith('llmc.')]:
                del sys.modules[k]
    raise RuntimeError('could not import the reference: ' + ', '.join(stubbed))


def run_one(M, arm, method, arch, assets, mdir, ddir, stubbed, ref_classes):
    """one pass of the reference's main(config) with the registry in the state `arm` asks for; returns the dump"""
    import numpy as np
    import torch
    from easydict import EasyDict
    from llmc.utils import check_config, seed_all
    from llmc.utils.registry_factory import ALGO_REGISTRY
    key = CONFIGS[method]['quant']['method']
    if arm == 'ours':
        sys.path.insert(0, ROOT)
        import llmc_amd
        llmc_amd.register_into(ALGO_REGISTRY)                     # <- the one line of INTEGRATION.md section 1
        assert ALGO_REGISTRY[key] is not ref_classes[key]
    else:
        for k, c in ref_classes.items():                          # the reference's own classes (undo an earlier 'ours' pass)
            ALGO_REGISTRY[k] = c
        if CONFIGS[method]['quant'].get('weight', {}).get('quant_type') == 'float-quant':
            bind_restated_qtorch()
    config = EasyDict(build_config(method, arch, mdir, ddir, os.path.join(assets, f'save_{arm}_{method}'), assets, arm))
    sp = config.quant.get('special', {})
    for k in ('scale_path', 'clip_path'):
        if sp.get(k):
            os.makedirs(sp[k], exist_ok=True)
    check_config(config)
    save_dir = None
    for flag, sub in (('save_vllm', 'vllm_quant_model'), ('save_autoawq', 'autoawq_quant_model')):
        if config.save.get(flag, False):           # the module-level global `if __name__ == '__main__'` sets (llmc/__main__.py:226-245)
            save_dir = os.path.join(config.save.save_path, sub)
            os.makedirs(save_dir, exist_ok=True)
            M.save_quant_path = save_dir
    seed_all(config.base.seed + 0)          # what `if __name__ == '__main__'` does before main() (llmc/__main__.py:179-300)

    # capture: the algorithm object main() builds, and what the reference's evaluator reports
    captured = {'ppl': []}
    cls = ALGO_REGISTRY[key]

    class Spy(cls):
        def __init__(self, *args, **kw):
            super().__init__(*args, **kw)
            captured['opt'] = self
    Spy.__name__ = cls.__name__
    ALGO_REGISTRY[key] = Spy
    import llmc.eval.utils as EU
    orig_eval = EU.eval_model

    def eval_spy(model, opts, eval_list, eval_pos):
        for ec, cfe in eval_list:
            if eval_pos in cfe.eval.eval_pos and not getattr(ec, '_spied', False):
                e0 = ec.eval

                def ev(model_, pos, _e0=e0):
                    r = _e0(model_, pos)
                    captured['ppl'].append((pos, float(r)))
                    return r
                ec.eval = ev
                ec._spied = True
        return orig_eval(model, opts, eval_list, eval_pos)
    M.eval_model = eval_spy
    try:
        M.main(config)
    finally:
        M.eval_model = orig_eval
        ALGO_REGISTRY[key] = cls
    opt = captured['opt']
    out = {'class_module': np.array(cls.__module__), 'stubbed': np.array(','.join(stubbed)),
           'ppl': np.array([p for _, p in captured['ppl']], dtype=np.float64)}
    model = opt.model.get_model() if hasattr(opt.model, 'get_model') else opt.model.model
    n = 0
    for name, mod in model.named_modules():
        w = getattr(mod, 'weight', None)
        if not (hasattr(mod, 'buf_scales') or type(mod).__name__.endswith('Linear')) or w is None or w.dim() != 2:
            continue
        if 'embed' in name or 'lm_head' in name:
            continue
        out[f'{name}/weight'] = w.detach().float().cpu().numpy()
        for b in ('buf_scales', 'buf_zeros', 'buf_perm', 'buf_qmax', 'buf_qmin', 'buf_act_scales_0'):
            t = getattr(mod, b, None)
            if torch.is_tensor(t):
                out[f'{name}/{b}'] = t.detach().float().cpu().numpy() if t.dtype != torch.int64 else t.cpu().numpy()
        out[f'{name}/type'] = np.array(type(mod).__module__ + '.' + type(mod).__name__)
        n += 1
    out['n_linear'] = np.array(n)
    for name, mod in model.named_modules():                        # clip_version v2 leaves its result in these buffers
        for b in ('buf_upbound_factor', 'buf_lowbound_factor'):
            t = getattr(mod, b, None)
            if torch.is_tensor(t):
                out[f'{name}/{b}'] = t.detach().float().cpu().numpy()
    if sp.get('scale_path') and os.path.exists(os.path.join(sp['scale_path'], 'scales.pth')):
        for k, v in torch.load(os.path.join(sp['scale_path'], 'scales.pth'), map_location='cpu').items():
            out['saved_scale/' + k] = v.float().numpy()
    if sp.get('clip_path') and os.path.exists(os.path.join(sp['clip_path'], 'clips.pth')):
        for bi, d in torch.load(os.path.join(sp['clip_path'], 'clips.pth'), map_location='cpu').items():
            for k, v in d.items():
                if torch.is_tensor(v):
                    out[f'saved_clip/{bi}/{k}'] = v.float().numpy()
    if save_dir:                                                    # the exported checkpoint, tensor by tensor, and its quantization config
        from safetensors.torch import load_file
        for f in sorted(os.listdir(save_dir)):
            if f.endswith('.safetensors'):
                for k, v in load_file(os.path.join(save_dir, f)).items():
                    if 'layers.' in k:
                        out['ckpt/' + k] = v.float().numpy() if v.is_floating_point() else v.numpy()
        cj = json.load(open(os.path.join(save_dir, 'config.json')))
        out['ckpt_config'] = np.array(json.dumps(cj.get('quantization_config', cj.get('compression_config', {})), sort_keys=True))
    print(f'ref_pipeline {arm} {method} {arch}: {n} linear layers, class from {cls.__module__}, ppl {captured["ppl"]}', flush=True)
    del opt, model, captured
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arms', default='ref', help='comma list of ref / ours, run in this order in ONE process')
    ap.add_argument('--methods', default='gptq', help='comma list of ' + ' / '.join(CONFIGS))
    ap.add_argument('--arch', default='llama', choices=['llama', 'opt'])
    ap.add_argument('--assets', required=True)
    ap.add_argument('--outdir', required=True, help='writes <outdir>/<arm>_<method>_<arch>.npz')
    ap.add_argument('--layers', type=int, default=2)
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    gpu = torch.cuda.is_available()
    ref_dir = os.path.join(ROOT, 'oracle', '_ref_gpu' if gpu else '_ref')
    if not os.path.isdir(os.path.join(ref_dir, 'llmc')):
        raise SystemExit(f'{ref_dir} missing (built by __graft_entry__.build() where /root/reference exists)')
    arms, methods = a.arms.split(','), a.methods.split(',')
    if 'ours' in arms and not gpu:
        raise SystemExit('--arms ours needs an MI355X (llmc_amd has no CPU path)')
    for k, v in (('RANK', '0'), ('LOCAL_RANK', '0'), ('WORLD_SIZE', '1'), ('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29611')):
        os.environ.setdefault(k, v)
    mdir, ddir = make_assets(a.assets, a.arch, n_layers=a.layers)
    M, stubbed = import_reference_main(ref_dir)
    from llmc.utils.registry_factory import ALGO_REGISTRY
    ref_classes = {k: ALGO_REGISTRY[k] for k in ('GPTQ', 'Awq', 'RTN', 'SpQR')}
    dist.init_process_group(backend='nccl' if gpu else 'gloo', rank=0, world_size=1)     # llmc/__main__.py:191
    if gpu:
        torch.cuda.set_device(0)
    os.makedirs(a.outdir, exist_ok=True)
    for method in methods:
        for arm in arms:
            out = run_one(M, arm, method, a.arch, a.assets, mdir, ddir, stubbed, ref_classes)
            np.savez(os.path.join(a.outdir, f'{arm}_{method}_{a.arch}.npz'), **out)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
