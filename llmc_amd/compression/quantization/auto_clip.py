"""AutoClipper with llmc's surface (llmc/compression/quantization/auto_clip.py:22-281), weight-only: clip_version v1
(search + clamp the weights to the searched range); of v2 the part that outlives the search: `apply_clip` / `get_clip_factor`
store a range as logit factors (`buf_upbound_factor` / `buf_lowbound_factor`) for a quantizer with `calib_algo: learnable`."""
import os

import torch
import torch.distributed as dist

from . import awq_ops
from .module_utils import _LLMC_LINEAR_TYPES_, _TRANSFORMERS_LINEAR_TYPES_


class AutoClipper:
    def __init__(self, w_only, wquantizer, aquantizer, clip_version, clip_sym, save_clip, padding_mask,
                 external_ranges=False):
        if clip_version not in ('v1', 'v2'):
            raise Exception('Not support other clip version')
        if clip_version == 'v2' and not external_ranges:
            # fail where the configuration is read, not after block 0's calibration forward: the v2 range SEARCH
            # (auto_clip.py:262-272) is not built — see _auto_clip_layer_v2. Callers that bring their own ranges (llmc's
            # two-stage pipelines load clips.pth) ask for the durable half explicitly: apply_clip / get_clip_factor.
            raise NotImplementedError('AutoClipper clip_version v2: the range search is outside the hot path (per_channel + '
                                      'activation-quantized pipelines). Pass external_ranges=True (quant.special.'
                                      'clip_external_ranges: True) to use apply_clip / get_clip_factor with ranges of your own')
        if not w_only:
            raise NotImplementedError('AutoClipper with activation quantization (fake_quantize_input, auto_clip.py:276-281) '
                                      'is outside the hot path')
        self.wquantizer = wquantizer
        self.aquantizer = aquantizer
        self.clip_version = clip_version
        self.clip_sym = clip_sym
        self.save_clip = save_clip
        self.padding_mask = padding_mask
        self.weight_clips = {}
        self.w_only = w_only
        self.logit = lambda x: torch.log(x / (1 - x))

    @torch.no_grad()
    def run(self, block, block_idx, input_feat, n_sample_token):
        for n, m in block.named_modules():
            if not isinstance(m, tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_)):
                continue
            if any(k in n for k in ['q_', 'k_', 'query', 'key', 'Wqkv']):      # auto_clip.py:56-60
                if self.clip_version == 'v2':
                    m.register_buffer('buf_upbound_factor', None)
                    m.register_buffer('buf_lowbound_factor', None)
                continue
            inputs = [torch.cat(input_feat[n])] if len(input_feat[n]) != 1 else input_feat[n]
            max_val, min_val = self.auto_clip_layer(block_idx, n, m.weight, inputs, n_sample_token=n_sample_token)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                for t in (max_val, min_val):                                    # auto_clip.py:72-76
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    t /= dist.get_world_size()
            self.apply_clip(block_idx, m, min_val, max_val, n)

    def _sample_tokens(self, x, i, w, n_sample_token):
        """auto_clip.py:133-147: flatten, drop padded tokens, every step-th token."""
        x = x.to(w.device)
        x = x.reshape(-1, x.shape[-1])
        if self.padding_mask and self.padding_mask[i].numel() == x.shape[0]:
            x = x[self.padding_mask[i].flatten().bool()]
        if n_sample_token is None:
            n_sample_token = min(x.shape[0], 512)
        step = max(1, x.shape[0] // n_sample_token)
        return x[0::step].contiguous(), n_sample_token

    @torch.no_grad()
    def auto_clip_layer(self, block_idx, layer_name, w, inputs, n_grid=20, max_shrink=0.5, n_sample_token=512,
                        eps=0.0):
        assert w.dim() == 2
        if self.clip_version == 'v2':
            return self._auto_clip_layer_v2(w, inputs, n_grid, max_shrink, n_sample_token)
        if len(inputs) == 1:                    # what run() always passes (it concatenates the batches, auto_clip.py:63-67)
            x, _ = self._sample_tokens(inputs[0], 0, w, n_sample_token)
            return awq_ops.clip_search(w.data, x, self.wquantizer, self.clip_sym, n_grid, max_shrink)
        # several batches (auto_clip.py:130-184): per shrink level err_mean = sum_i err_i / len(inputs) in the model
        # dtype, strict-< argmin in shrink order; n_sample_token, once derived from the first batch, is kept (:144-145)
        errs = None
        for i in range(len(inputs)):
            x, n_sample_token = self._sample_tokens(inputs[i], i, w, n_sample_token)
            e = awq_ops.clip_errs(w.data, x, self.wquantizer, self.clip_sym, n_grid, max_shrink)
            errs = e if errs is None else errs.add_(e)
        errs /= len(inputs)
        R, K = w.shape
        g = self.wquantizer.group_size if self.wquantizer.granularity == 'per_group' else K
        wg = w.data.reshape(R, K // g, g)
        org_max = (wg.abs() if self.clip_sym else wg).amax(dim=-1, keepdim=True)
        org_min = wg.amin(dim=-1, keepdim=True)
        return self._argmin_levels(errs, org_max, org_min, n_grid)

    def _levels(self, org_max, org_min, i_s, n_grid):
        # `tensor * python_float` as ATen's CPU kernels evaluate it: the product in fp32 (rounded), THEN the cast
        f = 1 - i_s / n_grid
        max_val = (org_max.float() * f).to(org_max.dtype)
        min_val = -max_val if self.clip_sym else (org_min.float() * f).to(org_min.dtype)
        return max_val, min_val

    def _argmin_levels(self, errs, org_max, org_min, n_grid):
        best_max, best_min = org_max.clone(), org_min.clone()
        min_errs = torch.ones_like(org_max) * 1e9
        for i_s in range(errs.shape[0]):
            max_val, min_val = self._levels(org_max, org_min, i_s, n_grid)
            err = errs[i_s].unsqueeze(-1)
            better = err < min_errs
            min_errs = torch.where(better, err, min_errs)
            best_max = torch.where(better, max_val, best_max)
            best_min = torch.where(better, min_val, best_min)
        return best_max, best_min

    def _auto_clip_layer_v2(self, w, inputs, n_grid, max_shrink, n_sample_token):
        """The v2 SEARCH (fake_quantize_weight's v2 branch, auto_clip.py:262-272) is not built: in the reference it only
        runs with per_channel weights — with per_group it raises inside its own quantizer (the [oc, 1, ng, 1] scales do not
        broadcast against the [-1, g] view, quant.py:701; reproduced by oracle/make_golden.py) — and every shipped config
        that selects it (awq_comb_omni/*/step_1_awq.yml, tesseraq_w4a16.yml) is per_channel with activation quantization
        or TesseraQ: outside the W4A16 hot path, and a 4096-wide group is beyond the clip kernel. What v2 leaves behind IS
        supported: apply_clip / get_clip_factor store the factors, the learnable-range quantizer and w_qdq consume them."""
        raise NotImplementedError('AutoClipper clip_version v2: the range search is outside the hot path (per_channel + '
                                  'activation-quantized pipelines); apply_clip / get_clip_factor and the learnable-range '
                                  'quantizer are available')

    def get_clip_factor(self, block_idx, layer, min_val, max_val, layer_name):
        """auto_clip.py:233-256."""
        t = self.wquantizer.reshape_tensor(layer.weight.data)
        org_min_val, org_max_val = t.amin(dim=-1, keepdim=True), t.amax(dim=-1, keepdim=True)
        org_val_shape = org_max_val.shape
        if self.clip_sym:
            abs_max_val = torch.max(org_max_val.abs(), org_min_val.abs()).clamp(min=1e-5)
            abs_max_val = abs_max_val.reshape(*max_val.shape[:2], -1)
            up_factor = self.logit(max_val / abs_max_val).reshape(org_val_shape)
            low_factor = None
        else:
            up_factor = self.logit(max_val / org_max_val.reshape(*max_val.shape[:2], -1)).reshape(org_val_shape)
            low_factor = self.logit(min_val / org_min_val.reshape(*min_val.shape[:2], -1)).reshape(org_val_shape)
        return up_factor, low_factor

    @torch.no_grad()
    def apply_clip(self, block_idx, layer, min_val, max_val, layer_name):
        if self.clip_version == 'v2':                 # auto_clip.py:213-229
            up_factor, low_factor = self.get_clip_factor(block_idx, layer, min_val, max_val, layer_name)
            layer.register_buffer('buf_upbound_factor', up_factor)
            layer.register_buffer('buf_lowbound_factor', low_factor)
            if self.save_clip:
                n = f'{layer_name}.weight_quantizer.'
                self.weight_clips.setdefault(block_idx, {})[n + 'upbound_factor'] = up_factor.cpu()
                self.weight_clips[block_idx][n + 'lowbound_factor'] = None if low_factor is None else low_factor.cpu()
            return
        g = self.wquantizer.group_size if self.wquantizer.granularity == 'per_group' else layer.weight.shape[1]
        if self.clip_sym:
            min_val = -max_val
        w = layer.weight.data.contiguous()
        awq_ops.clamp_groups_(w, min_val.reshape(-1), max_val.reshape(-1), g)
        layer.weight.data = w
