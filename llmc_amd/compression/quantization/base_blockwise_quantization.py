"""BaseBlockwiseQuantization with llmc's operator surface (llmc/compression/quantization/
base_blockwise_quantization.py:41-1038): same constructor, overridables, buffer names and config keys, so the
subclasses registered in ALGO_REGISTRY ('GPTQ', 'Awq', 'RTN') drop into llmc's `__main__.main`.

In scope: quantizer selection, collect_block_qparams, the block / subset loop with
true_sequential re-hooking and quant_out, apply_scale (LN->fc and fc->fc), scaling_input/update_input_feat,
static per-tensor activation qparams, deploy to fake / real-quant wrappers. Out of scope and rejected loudly:
rotations (QuaRot), KV-cache quantization, quantized attention / act-fn modules, token reduction, FP8
block-wise checkpoints (DeepSeek). Mixed precision (`ignored_layers`) is in since round 4.
"""
import copy
import functools
import gc
import os
from collections import defaultdict
from functools import partial

import torch
import torch.distributed as dist

from ..blockwise_optimization import BlockwiseOpt
from . import awq_ops
from .module_utils import (_LLMC_LINEAR_TYPES_, _LLMC_LN_TYPES_, _REALQUANT_LINEAR_MAP_,
                           _TRANSFORMERS_LINEAR_TYPES_, _TRANSFORMERS_LN_TYPES_, EffcientFakeQuantLinear,
                           FakeQuantLinear, OriginFloatLinear)
from .quant import FloatQuantizer, IntegerQuantizer


def _get(cfg, key, default=None):
    """dict / EasyDict / attribute access alike."""
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class BaseBlockwiseQuantization(BlockwiseOpt):
    def __init__(self, model, quant_config, input, padding_mask, config):
        super().__init__(model, quant_config, input, padding_mask, config)
        self.dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.set_quant_config()

    # ---- block-wise FP8 checkpoints (DeepSeek-V3 layout: `weight` float8_e4m3fn + `weight_scale_inv`) ------------
    def _fp8_to_bf16(self, weight, scale_inv):
        """weight_cast_to_bf16(...).to(bfloat16) of the reference's FP8 branches (base_…:53-57, 347-350, 666-670, 763-767;
        awq.py:53-56, 148-151). The reference binds the Triton kernels on FP8-capable GPUs and the FloatQuantizer spelling
        elsewhere (awq.py:14-20); gfx950 is FP8-capable, so the kernel arithmetic is the default here and
        `special.fp8_cast: quantizer` selects the other one (its rounding then follows the quantizer's fp8_semantics)."""
        from . import kernel, quant
        if self.fp8_cast == 'quantizer':
            return quant.weight_cast_to_bf16(weight.data, scale_inv.data, self.fp8_block_size)
        return kernel.weight_cast_to_bf16(weight.data.contiguous(), scale_inv.data.contiguous(), self.fp8_block_size)

    def _bf16_to_fp8(self, weight):
        """weight_cast_to_fp8(weight, self.fp8_block_size) -> (float8_e4m3fn weight, fp32 block scales)."""
        from . import kernel, quant
        if self.fp8_cast == 'quantizer':
            return quant.weight_cast_to_fp8(weight, self.fp8_block_size, fp8_semantics=self.fp8_cast_semantics)
        return kernel.weight_cast_to_fp8(weight.contiguous(), self.fp8_block_size)

    @staticmethod
    def _is_fp8(module):
        return module.weight.data.dtype == torch.float8_e4m3fn

    # ---- quantizer callbacks handed to the Linear wrappers (base_…:46-83) ---------------------------
    def w_qdq(self, module, wquantizer):
        args = {}                                 # base_blockwise_quantization.py:46-52: clip v2's learnable bounds ride along
        if getattr(module, 'buf_upbound_factor', None) is not None:
            args['upbound_factor'] = module.buf_upbound_factor
            args['lowbound_factor'] = getattr(module, 'buf_lowbound_factor', None)
        if self._is_fp8(module):                  # base_…:53-67: de-block, fake-quantize, re-block (the scales are replaced)
            tmp = wquantizer.fake_quant_weight_dynamic(self._fp8_to_bf16(module.weight, module.weight_scale_inv), args)
            tmp, module.weight_scale_inv.data = self._bf16_to_fp8(tmp)
            return tmp
        return wquantizer.fake_quant_weight_dynamic(module.weight, args)

    def w_q(self, module, wquantizer):
        return wquantizer.real_quant_weight_dynamic(module.weight.data)

    def a_qdq(self, act, module, aquantizer, input_index=0):
        if self.act_static:
            args = {k: getattr(module, f'buf_act_{k}_{input_index}', None) for k in ('scales', 'zeros', 'qmax', 'qmin')}
            return aquantizer.fake_quant_act_static(act, args)
        return aquantizer.fake_quant_act_dynamic(act)

    def get_replacement_params(self, mode='fake_quant', w_only=False, name=None):
        if mode in ('fake_quant', 'fake_quant_wo_kv'):
            return {'a_qdq': partial(self.a_qdq, aquantizer=self.aquantizer) if not w_only else None,
                    'w_qdq': partial(self.w_qdq, wquantizer=self.wquantizer)}
        if mode in _REALQUANT_LINEAR_MAP_:
            return {'w_q': partial(self.w_q, wquantizer=self.wquantizer), 'quant_config': self.quant_config}
        if mode == 'origin_float':
            return {}
        raise NotImplementedError(f'replacement mode {mode} is outside the hot path')

    # ---- configuration (base_…:133-300) ------------------------------------------------------------
    def set_quant_config(self):
        qc = self.quant_config
        # mixed precision (base_blockwise_quantization.py:137-144): layers named here stay in floating point at deploy time
        # (set_no_quant_layer marks them `no_quant`, the model adapter's replace_module_subset skips marked modules,
        # models/base_model.py:433-435). They are still calibrated / transformed like the reference does.
        if 'ignored_layers' in self.config:
            il = self.config['ignored_layers']
            self.mixed_precision = True
            self.ignored_block_ids = _get(il, 'block_ids', []) or []
            self.ignored_layer_names = _get(il, 'layer_names', []) or []
            self.ignored_speical_names = _get(il, 'speical_names', []) or []      # (spelling as in the reference)
        else:
            self.mixed_precision = False
        self.quant_out = _get(qc, 'quant_out', False)
        self.tp = _get(qc, 'tp', 1)

        def make(cfg):
            cfg = dict(cfg)
            qt = cfg.pop('quant_type', 'int-quant')
            if qt == 'int-quant':
                if cfg.get('bit') == 48:
                    raise NotImplementedError('W48 quantizer is outside the hot path')
                return IntegerQuantizer(**cfg)
            if qt == 'float-quant':
                return FloatQuantizer(**cfg)
            raise ValueError(f'unknown quant_type {qt}')

        self.wquantizer = make(qc['weight'])
        if 'act' in qc:
            self.w_only = False
            self.aquantizer = make(qc['act'])
            self.act_static = _get(qc['act'], 'static', False)
            if self.act_static:
                assert qc['act']['granularity'] == 'per_tensor', 'Only support per_tensor static quant'
            if _get(qc['act'], 'quant_attn', False) or _get(qc['act'], 'quant_act_fn', False):
                raise NotImplementedError('quantized attention / activation functions are outside the hot path')
        else:
            self.w_only = True
            self.aquantizer = None
            self.act_static = False
        self.quant_attn = self.quant_softmax = self.quant_act_fn = False
        if 'kvcache' in qc:
            raise NotImplementedError('KV-cache quantization is outside the hot path')
        self.quant_kvcache = False

        special = _get(qc, 'special', {}) or {}
        self.true_sequential = special.get('true_sequential', False)
        self.weight_clip = special.get('weight_clip', False)
        if self.weight_clip:
            from .auto_clip import AutoClipper
            self.save_clip = special.get('save_clip', False)
            if self.save_clip:
                self.clip_path = special['clip_path']
            self.clip_version = special.get('clip_version', 'v1')
            if self.clip_version == 'v2':
                assert self.wquantizer.calib_algo == 'learnable'      # base_blockwise_quantization.py:229-230
            self.auto_clipper = AutoClipper(
                w_only=self.w_only, wquantizer=self.wquantizer, aquantizer=self.aquantizer,
                clip_version=self.clip_version, clip_sym=special.get('clip_sym', self.wquantizer.sym),
                save_clip=self.save_clip, padding_mask=self.padding_mask,
                external_ranges=special.get('clip_external_ranges', False))
        self.save_scale = special.get('save_scale', False)
        if self.save_scale:
            self.scale_path = special['scale_path']
            self.act_scales = {}
        if special.get('online_rotate', False):
            raise NotImplementedError('online rotation is outside the hot path')
        self.online_rotate = False
        self.modality = _get(qc, 'modality', 'language')
        # base_…:134-135: block size of a block-wise FP8 checkpoint (model.fp8_block_size; DeepSeek-V3: 128)
        self.fp8_block_size = getattr(self.model, 'fp8_block_size', None) or 128
        self.fp8_cast = special.get('fp8_cast', 'kernel')
        if self.fp8_cast not in ('kernel', 'quantizer'):
            raise ValueError(f"special.fp8_cast must be 'kernel' or 'quantizer', got {self.fp8_cast!r}")
        self.fp8_cast_semantics = special.get('fp8_cast_semantics', 'qtorch')
        if getattr(self, 'auto_clipper', None) is not None:       # auto_clip.py:47-53, 78-81 use the same casts
            self.auto_clipper.fp8_block_size = self.fp8_block_size
            self.auto_clipper.fp8_to_bf16 = self._fp8_to_bf16
            self.auto_clipper.bf16_to_fp8 = self._bf16_to_fp8
        self.do_gqa_trans = special.get('do_gqa_trans', False)
        self.set_model_config()

    def set_model_config(self):
        mc = getattr(self.model, 'model_config', None)
        self.has_gqa = False
        if mc is None:
            return
        self.hidden_size = getattr(mc, 'hidden_size', None)
        self.num_heads = getattr(mc, 'num_attention_heads', None)
        if self.hidden_size and self.num_heads:
            self.head_dim = self.hidden_size // self.num_heads
        if getattr(mc, 'num_key_value_heads', None):
            self.num_key_value_heads = mc.num_key_value_heads
            self.num_key_value_groups = self.num_heads // self.num_key_value_heads
            self.has_gqa = self.num_key_value_groups > 1

    # ---- RTN qparams of every Linear (base_…:338-365) -------------------------------------------------
    @torch.no_grad()
    def collect_block_qparams(self, block):
        for n, m in self.model.get_block_linears(block).items():
            _, scales, zeros, max_int, min_int = self.wquantizer.get_tensor_qparams(m.weight.data)
            m.register_buffer('buf_scales', scales.detach())
            m.register_buffer('buf_zeros', zeros.detach())
            m.register_buffer('buf_qmax', torch.as_tensor(max_int).to(m.weight.device))
            m.register_buffer('buf_qmin', torch.as_tensor(min_int).to(m.weight.device))

    # ---- block loop (base_…:367-526) ----------------------------------------------------------------
    def block_forward(self, block, input_data=None):
        output = []
        if input_data is None:
            input_data = self.input['data']
        dev = next(block.parameters()).device
        for i in range(len(input_data)):
            input_data[i] = input_data[i].to(device=dev)
            kw = self.input['kwargs'][i]
            for k, v in kw.items():
                if torch.is_tensor(v):
                    kw[k] = v.to(device=dev)
                elif isinstance(v, tuple):
                    kw[k] = tuple(t.to(device=dev) if torch.is_tensor(t) else t for t in v)
            # one forward pass of the block = one calibration sample: hooks that dedupe work across the layers of a
            # subset (GPTQ's shared Hessians) key on this counter
            self._fwd_pass = getattr(self, '_fwd_pass', 0) + 1
            with torch.no_grad():
                out = block(input_data[i], **kw)
            output.append(out[0] if isinstance(out, tuple) else out)
        return output

    def block_opt(self, block):
        block = block.cuda()
        named_linears = self.model.get_block_linears(block)
        extra_modules = self.model.get_extra_modules(block) if hasattr(self.model, 'get_extra_modules') else {}
        modules = {**named_linears, **extra_modules}
        input_feat = defaultdict(list)
        handles = self.register_hooks(modules, input_feat)
        self.block_init(block)
        self.run(block, input_feat, handles)
        block = block.cpu()
        del input_feat, block
        gc.collect()
        torch.cuda.empty_cache()

    def register_hooks(self, input_feat_modules, input_feat):
        handles = []
        if not self.data_free:
            for name, mod in input_feat_modules.items():
                handles.append(mod.register_forward_hook(
                    functools.partial(self.cache_input_hook, name=name, feat_dict=input_feat)))
        return handles

    def run(self, block, input_feat, handles):
        if not self.data_free:
            if self.quant_out:
                self.block_forward(block)
            else:
                self.input['data'] = self.block_forward(block)
            for h in handles:
                h.remove()
            self.block_transform(block, input_feat, self.input['kwargs'])
        else:
            self.block_transform(block)
        if not self.data_free and self.quant_out:
            self.model.replace_module_block(FakeQuantLinear, block, self.block_idx,
                                            self.get_replacement_params('fake_quant', self.w_only))
            self.input['data'] = self.block_forward(block)

    def block_transform(self, block, input_feat, block_kwargs):
        subsets = self.model.get_subsets_in_block(block)
        for index, subset in enumerate(subsets):
            layers_dict = subset['layers']
            input_name = subset['input'][0]
            if subset['has_kwargs']:
                if 'sub_keys' in subset:
                    subset_kwargs = [{k: kw[v] for k, v in subset['sub_keys'].items()} for kw in block_kwargs]
                else:
                    subset_kwargs = block_kwargs
            else:
                subset_kwargs = {}
            self.subset_transform(subset, input_feat, subset_kwargs)
            if self.act_static:
                self.register_act_qparams(layers_dict, copy.copy(input_feat[input_name]))
            if self.true_sequential and index != len(subsets) - 1:
                input_feat.update(self.rehook_next_subset(block, subset, subsets[index + 1]))

    def rehook_next_subset(self, block, subset, next_subset):
        self.subset_init(next_subset)
        self.model.replace_module_subset(FakeQuantLinear, block, subset, self.block_idx,
                                         self.get_replacement_params('fake_quant', self.w_only))
        feat = defaultdict(list)
        handles = self.register_hooks(next_subset['layers'], feat)
        self.block_forward(block)
        for h in handles:
            h.remove()
        return feat

    # ---- static activation qparams (base_…:567-588): ranges from the quantizer (quant.py:561-586), one entry per input ----
    @torch.no_grad()
    def register_act_qparams(self, layers_dict, act_tensors):
        scales_list, zeros_list, qmin_list, qmax_list = self.aquantizer.get_batch_tensors_qparams(act_tensors)
        for i, (scales, zeros, qmin, qmax) in enumerate(zip(scales_list, zeros_list, qmin_list, qmax_list)):
            dev = scales.device
            if _world() > 1:
                dist.all_reduce(scales, op=dist.ReduceOp.SUM)
                scales = scales / _world()
            for layer in layers_dict.values():
                if isinstance(layer, tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_)):
                    layer.register_buffer(f'buf_act_scales_{i}', scales)
                    layer.register_buffer(f'buf_act_zeros_{i}', zeros.to(dev))
                    layer.register_buffer(f'buf_act_qmin_{i}', qmin.to(dev))
                    layer.register_buffer(f'buf_act_qmax_{i}', qmax.to(dev))

    # ---- scale folding (base_…:597-611, 632-700, 750-778) ---------------------------------------------
    @torch.no_grad()
    def repeat_gqa_scales(self, scales):
        scales = scales.view(1, self.num_key_value_heads, self.head_dim)
        return torch.repeat_interleave(scales, dim=1, repeats=self.num_key_value_groups)

    @torch.no_grad()
    def apply_scale(self, scales, prev_op, layers):
        assert len(prev_op) == 1, 'Only support single prev_op. If multi prev_ops, code need to be updated.'
        if isinstance(prev_op[0], tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_)):
            assert len(layers) == 1
            self.scale_fc_fc(prev_op[0], layers[0], scales)
        elif isinstance(prev_op[0], tuple(_LLMC_LN_TYPES_ + _TRANSFORMERS_LN_TYPES_)) or hasattr(prev_op[0], 'weight'):
            self.scale_ln_fcs(prev_op[0], layers, scales)
        else:
            raise NotImplementedError(f'prev_op {type(prev_op[0])} not supported yet!')

    @torch.no_grad()
    def scale_fc_fc(self, fc1, fc2, scales):
        scales = scales.to(fc1.weight.device)
        if fc1.out_features == fc2.in_features * 2:                    # fused gate|up: scale the second half
            half = fc1.weight.shape[0] // 2
            fc1.weight.data[half:].div_(scales.view(-1, 1))
            if getattr(fc1, 'bias', None) is not None:
                fc1.bias.data[half:].div_(scales.view(-1))
        elif fc1.out_features == fc2.in_features:
            if getattr(fc1, 'bias', None) is not None:
                fc1.bias.div_(scales.view(-1))
            if self._is_fp8(fc1):                                      # base_…:666-674
                tmp = self._fp8_to_bf16(fc1.weight, fc1.weight_scale_inv)
                tmp.div_(scales.view(-1, 1))
                fc1.weight.data, fc1.weight_scale_inv.data = self._bf16_to_fp8(tmp)
            else:
                fc1.weight.div_(scales.view(-1, 1))
        elif self.has_gqa and self.do_gqa_trans:
            if getattr(fc1, 'bias', None) is not None:
                fc1.bias.div_(scales.view(-1))
            fc1.weight.div_(scales.view(-1, 1))
            scales = self.repeat_gqa_scales(scales).reshape(-1)
        else:
            raise Exception('Can not scale this fc-fc.')
        self._mul_cols_weight(fc2, scales.reshape(-1))

    def _mul_cols_weight(self, fc, cols):
        """fc.weight.mul_(scales.view(1, -1)); a block-wise FP8 weight is de-blocked to bf16 first and re-blocked after
        (base_…:691-700, 763-772)."""
        if self._is_fp8(fc):
            tmp = self._fp8_to_bf16(fc.weight, fc.weight_scale_inv)
            awq_ops.mul_cols_(tmp, cols.to(tmp.dtype))
            fc.weight.data, fc.weight_scale_inv.data = self._bf16_to_fp8(tmp)
        else:
            awq_ops.mul_cols_(fc.weight.data, cols.to(fc.weight.dtype))

    @torch.no_grad()
    def scale_ln_fcs(self, ln, fcs, scales):
        if not isinstance(fcs, list):
            fcs = [fcs]
        scales = scales.to(ln.weight.device).to(ln.weight.dtype)
        ln.weight.div_(scales)
        if getattr(ln, 'bias', None) is not None:
            ln.bias.div_(scales)
        for fc in fcs:
            self._mul_cols_weight(fc, scales.reshape(-1))
        for p in list(ln.parameters()) + [q for fc in fcs for q in fc.parameters()]:
            # (isnan has no float8 kernel: an e4m3fn NaN is the code 0x7f / 0xff)
            nan = ((p.view(torch.uint8) & 0x7f) == 0x7f) if p.dtype == torch.float8_e4m3fn else torch.isnan(p)
            assert nan.sum() == 0

    @torch.no_grad()
    def scaling_input(self, x, scales, is_gqa):
        s = self.repeat_gqa_scales(scales).reshape(-1) if is_gqa else scales.reshape(-1)
        return awq_ops.div_cols(x, s)

    @torch.no_grad()
    def update_input_feat(self, scale, input_feat, layers_dict, is_gqa):
        done = {}
        for name in layers_dict:
            for i, inp in enumerate(input_feat[name]):
                key = (inp.data_ptr(), tuple(inp.shape))
                if key not in done:                                    # layers of a subset share the tensor
                    done[key] = self.scaling_input(inp, scale.to(inp.device), is_gqa)
                input_feat[name][i] = done[key]

    # ---- deploy / save (base_…:933-1038) ---------------------------------------------------------------
    def set_no_quant_layer(self):
        """base_blockwise_quantization.py:910-932: `block_ids` (ints or 'a-b' ranges) x `layer_names` inside a block, or full
        names `<block_name_prefix>.<idx>.<name>` in `speical_names`, get a `no_quant` buffer."""
        import re
        if self.ignored_speical_names:
            assert hasattr(self.model, 'block_name_prefix'), 'block_name_prefix missing in model'
        ids = []
        for item in self.ignored_block_ids:
            m = re.match(r'(\d+)-(\d+)', str(item))
            if m:
                ids.extend(range(int(m.group(1)), int(m.group(2)) + 1))
            else:
                ids.append(int(item))
        for idx, block in enumerate(self.blocks):
            for n, m in block.named_modules():
                if idx in ids and n in self.ignored_layer_names:
                    m.register_buffer('no_quant', torch.tensor(True))
                elif self.ignored_speical_names and f'{self.model.block_name_prefix}.{idx}.{n}' in self.ignored_speical_names:
                    m.register_buffer('no_quant', torch.tensor(True))

    @torch.no_grad()
    def deploy(self, quant_format, keep_device=False):
        mapping = {'origin_float': OriginFloatLinear, 'fake_quant': EffcientFakeQuantLinear,
                   'fake_quant_wo_kv': EffcientFakeQuantLinear}
        mapping.update(_REALQUANT_LINEAR_MAP_)
        if quant_format not in mapping:
            raise NotImplementedError(f"Quant format '{quant_format}' is not implemented.")
        if self.mixed_precision and 'quant' in quant_format:
            self.set_no_quant_layer()
        self.model.replace_language_module_all(mapping[quant_format],
                                               self.get_replacement_params(quant_format, self.w_only),
                                               keep_device=keep_device)

    @torch.no_grad()
    def save_model(self, path):
        """base_blockwise_quantization.py:1016-1038: rank 0 writes the (deployed) model + tokenizer. HF models go through
        their own `save_pretrained` (safetensors + config.json, the files vLLM / AutoAWQ load); any other module is
        written the same way — `model.safetensors` of the state dict and a `config.json` for the exporters to extend."""
        if int(os.environ.get('RANK', '0')) != 0:
            return
        for t in list(self.model.get_model().parameters()) + list(self.model.get_model().buffers()):
            if not t.is_contiguous():                          # contiguous_params (base_blockwise_quantization.py:996-1013)
                t.data = t.data.contiguous()
        net = self.model.get_model()
        if hasattr(net, 'save_pretrained'):
            net.save_pretrained(path)
        else:
            import json

            from safetensors.torch import save_file
            os.makedirs(path, exist_ok=True)
            sd = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items() if torch.is_tensor(v)}
            save_file(sd, os.path.join(path, 'model.safetensors'))
            mc = getattr(self.model, 'model_config', None)
            if mc is not None and hasattr(mc, 'to_dict'):
                cfg = dict(mc.to_dict())
            elif mc is not None and hasattr(mc, '__dict__'):
                cfg = dict(vars(mc))
            else:
                cfg = {}
            cfg = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, list, type(None)))}
            with open(os.path.join(path, 'config.json'), 'w') as f:
                json.dump(cfg, f, indent=4)
        if getattr(self.model, 'tokenizer', None) is not None:
            self.model.tokenizer.save_pretrained(path)
