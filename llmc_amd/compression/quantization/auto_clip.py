"""AutoClipper with llmc's surface (llmc/compression/quantization/auto_clip.py:22-281), clip_version v1, w_only."""
import os

import torch
import torch.distributed as dist

from . import awq_ops
from .module_utils import _LLMC_LINEAR_TYPES_, _TRANSFORMERS_LINEAR_TYPES_


class AutoClipper:
    def __init__(self, w_only, wquantizer, aquantizer, clip_version, clip_sym, save_clip, padding_mask):
        if clip_version != 'v1' or not w_only:
            raise NotImplementedError('AutoClipper: only clip_version v1 with weight-only quantization is on the '
                                      'accelerated path')
        self.wquantizer = wquantizer
        self.aquantizer = aquantizer
        self.clip_version = clip_version
        self.clip_sym = clip_sym
        self.save_clip = save_clip
        self.padding_mask = padding_mask
        self.weight_clips = {}
        self.w_only = w_only

    @torch.no_grad()
    def run(self, block, block_idx, input_feat, n_sample_token):
        for n, m in block.named_modules():
            if not isinstance(m, tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_)):
                continue
            if any(k in n for k in ['q_', 'k_', 'query', 'key', 'Wqkv']):      # auto_clip.py:56-60
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
        best_max, best_min = org_max.clone(), org_min.clone()
        min_errs = torch.ones_like(org_max) * 1e9
        for i_s in range(errs.shape[0]):
            # `tensor * python_float` as ATen's CPU kernels evaluate it: the product in fp32 (rounded), THEN the cast — two
            # separate device ops here, because a fused fp16 multiply rounds the exact product once and lands on the
            # other side of a tie now and then (llmc_amd/csrc/common.h, f32_to_f16_bits)
            f = 1 - i_s / n_grid
            max_val = (org_max.float() * f).to(org_max.dtype)
            min_val = -max_val if self.clip_sym else (org_min.float() * f).to(org_min.dtype)
            err = errs[i_s].unsqueeze(-1)
            better = err < min_errs
            min_errs = torch.where(better, err, min_errs)
            best_max = torch.where(better, max_val, best_max)
            best_min = torch.where(better, min_val, best_min)
        return best_max, best_min

    @torch.no_grad()
    def apply_clip(self, block_idx, layer, min_val, max_val, layer_name):
        g = self.wquantizer.group_size if self.wquantizer.granularity == 'per_group' else layer.weight.shape[1]
        if self.clip_sym:
            min_val = -max_val
        w = layer.weight.data.contiguous()
        awq_ops.clamp_groups_(w, min_val.reshape(-1), max_val.reshape(-1), g)
        layer.weight.data = w
