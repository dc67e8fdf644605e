"""AutoClipper with llmc's surface (llmc/compression/quantization/auto_clip.py:22-281): clip_version v1 (search + clamp the
weights to the searched range) and v2 (search over learnable-range factors; `apply_clip` / `get_clip_factor` store the range
as logit factors `buf_upbound_factor` / `buf_lowbound_factor` for a quantizer with `calib_algo: learnable`), weight-only or
with quantized activations. Two routes, same arithmetic:
  * W4A16-style (v1, weight-only, integer min/max quantizer, groups <= 128): one kernel builds the candidates of every
    shrink level, evaluates them and takes the argmin (llmc_awq_clip_search);
  * everything else (per_channel / per_tensor ranges, FP8 quantizers, quantized activations, v2): the candidates of every
    shrink level come from the quantizer's own kernels exactly as fake_quantize_weight forms them (auto_clip.py:258-274,
    in the reference's output-channel batches of 256 / 64 rows, which is what a per_tensor range spans), the error table
    from llmc_awq_clip_errs_cand, the strict-< argmin over the ten levels on [R, ng] tensors."""
import torch
import torch.distributed as dist

from . import awq_ops
from .module_utils import _LLMC_LINEAR_TYPES_, _TRANSFORMERS_LINEAR_TYPES_


class AutoClipper:
    def __init__(self, w_only, wquantizer, aquantizer, clip_version, clip_sym, save_clip, padding_mask,
                 external_ranges=False):
        if clip_version not in ('v1', 'v2'):
            raise Exception('Not support other clip version')
        if clip_version == 'v2' and not external_ranges and getattr(wquantizer, 'granularity', None) == 'per_group':
            # fail where the configuration is read, not after block 0's calibration forward: with per_group
            # weights the reference's v2 search raises inside its own quantizer (the [oc, 1, ng, 1] scales do not broadcast
            # against the [-1, g] view, quant.py:701; reproduced by oracle/make_golden.py). Callers that bring their own
            # ranges (llmc's two-stage pipelines load clips.pth) use apply_clip / get_clip_factor: external_ranges=True.
            raise NotImplementedError('AutoClipper clip_version v2 searches per_channel / per_tensor ranges only (per_group '
                                      'fails in the reference too). Pass external_ranges=True (quant.special.'
                                      'clip_external_ranges: True) to use apply_clip / get_clip_factor with ranges of your own')
        self.external_ranges = external_ranges
        self.wquantizer = wquantizer
        self.aquantizer = aquantizer
        self.clip_version = clip_version
        self.clip_sym = clip_sym
        self.save_clip = save_clip
        self.padding_mask = padding_mask
        self.weight_clips = {}
        self.w_only = w_only
        self.logit = lambda x: torch.log(x / (1 - x))
        # block-wise FP8 checkpoints: (weight, weight_scale_inv) -> bf16 and bf16 -> (fp8 weight, block scales); bound by the
        # owning algorithm to its own casts (base_blockwise_quantization.py: _fp8_to_bf16 / _bf16_to_fp8)
        self.fp8_block_size = 128
        self.fp8_to_bf16 = None
        self.bf16_to_fp8 = None

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
            # auto_clip.py:47-53, 78-81: a block-wise FP8 checkpoint weight (DeepSeek-V3 layout) is de-blocked to bf16 for the
            # search and the clamp, and re-blocked afterwards. (The reference reads `self.fp8_block_size` here, which its
            # AutoClipper never sets — as shipped it raises AttributeError on such checkpoints; the casts and the block size
            # are handed over by the owning algorithm: BaseBlockwiseQuantization.set_quant_config.)
            is_fp8_weight = m.weight.data.dtype == torch.float8_e4m3fn
            if is_fp8_weight:
                if self.fp8_to_bf16 is None:
                    raise RuntimeError('AutoClipper: block-wise FP8 weight but no cast bound (fp8_to_bf16 / bf16_to_fp8)')
                m.weight.data = self.fp8_to_bf16(m.weight, m.weight_scale_inv)
            inputs = [torch.cat(input_feat[n])] if len(input_feat[n]) != 1 else input_feat[n]
            max_val, min_val = self.auto_clip_layer(block_idx, n, m.weight, inputs, n_sample_token=n_sample_token)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                for t in (max_val, min_val):                                    # auto_clip.py:72-76
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    t /= dist.get_world_size()
            self.apply_clip(block_idx, m, min_val, max_val, n)
            if is_fp8_weight:
                m.weight.data, m.weight_scale_inv.data = self.bf16_to_fp8(m.weight.data)

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
        if not self._fused_route():
            return self._auto_clip_layer_general(w, inputs, n_grid, max_shrink, n_sample_token, eps)
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

    def _argmin_levels(self, errs, org_max, org_min, n_grid, levels=None):
        """The reference keeps, per group, the (max, min) pair of the level whose error is smallest — the pair it EVALUATED
        (auto_clip.py:153-176): `levels` are the shrink levels the candidates were built with (level 0 is 0 + eps for v2 with
        quantized activations, auto_clip.py:128-130), default 0, 1, 2, ..."""
        best_max, best_min = org_max.clone(), org_min.clone()
        min_errs = torch.ones_like(org_max) * 1e9
        for i_s in range(errs.shape[0]):
            max_val, min_val = self._levels(org_max, org_min, i_s if levels is None else levels[i_s], n_grid)
            err = errs[i_s].unsqueeze(-1)
            better = err < min_errs
            min_errs = torch.where(better, err, min_errs)
            best_max = torch.where(better, max_val, best_max)
            best_min = torch.where(better, min_val, best_min)
        return best_max, best_min

    def _fused_route(self):
        from .quant import IntegerQuantizer
        wq = self.wquantizer
        return (self.clip_version == 'v1' and self.w_only and isinstance(wq, IntegerQuantizer) and wq.calib_algo == 'minmax'
                and wq.round_zp and wq.granularity == 'per_group' and wq.group_size <= 128)

    def fake_quantize_weight(self, w, min_val, max_val, org_min_val, org_max_val):
        """auto_clip.py:258-274 for one output-channel batch: w [oc, K], ranges [oc, ng, 1]."""
        wq = self.wquantizer
        oc, K = w.shape
        if self.clip_version == 'v1':
            g = K // max_val.shape[1]
            return wq.fake_quant_weight_dynamic(awq_ops.clamp_groups_(w.clone(), min_val, max_val, g))
        if wq.granularity == 'per_group':
            raise NotImplementedError('AutoClipper clip_version v2 with per_group weights: the reference raises here too '
                                      '(quant.py:701: [oc, 1, ng, 1] scales against the [-1, g] view)')
        low_factor = self.logit(min_val / org_min_val)
        up_factor = self.logit(max_val / org_max_val)
        tensor_range = wq.get_learnable_range(w.reshape(oc, max_val.shape[1], -1), low_factor, up_factor)
        scales, zeros, qmax, qmin = wq.get_qparams(tensor_range, w.device)
        return wq.fake_quant_weight_static(w, {'scales': scales, 'zeros': zeros, 'qmax': qmax, 'qmin': qmin})

    def fake_quantize_input(self, block_idx, x, layer_name):
        """auto_clip.py:276-281; x is the [1, tok, ng, g] view the reference quantizes (a per_token range is per token AND
        group there)."""
        return x if self.w_only else self.aquantizer.fake_quant_act_dynamic(x)

    def _auto_clip_layer_general(self, w, inputs, n_grid, max_shrink, n_sample_token, eps=0.0):
        """auto_clip.py:84-191 for every quantizer / granularity / clip version (see the module docstring)."""
        wq = self.wquantizer
        wd = w.data.contiguous()
        R, K = wd.shape
        g = wq.group_size if wq.granularity == 'per_group' else K
        if K % g != 0:
            raise NotImplementedError('AutoClipper: group size must divide the row (the padded view of auto_clip.py:103-105 '
                                      'is outside the hot path)')
        ng = K // g
        oc = 256 if R % 256 == 0 else 64                                   # auto_clip.py:106-107
        assert R % oc == 0
        wg = wd.reshape(R, ng, g)
        org_max = (wg.abs() if self.clip_sym else wg).amax(dim=-1, keepdim=True)
        org_min = wg.amin(dim=-1, keepdim=True)
        ns = int(max_shrink * n_grid)
        cands = torch.empty((ns, R, K), dtype=wd.dtype, device=wd.device)
        levels = [i_s + eps if (i_s == 0 and self.clip_version == 'v2' and not self.w_only) else i_s for i_s in range(ns)]   # :128-130
        for i_s in range(ns):
            max_val, min_val = self._levels(org_max, org_min, levels[i_s], n_grid)
            for b0 in range(0, R, oc):
                sl = slice(b0, b0 + oc)
                cands[i_s, sl] = self.fake_quantize_weight(wd[sl], min_val[sl], max_val[sl], org_min[sl], org_max[sl])
        errs = None
        for i in range(len(inputs)):
            x, n_sample_token = self._sample_tokens(inputs[i], i, w, n_sample_token)
            xq = None
            if not self.w_only:
                xq = self.fake_quantize_input(None, x.reshape(1, x.shape[0], ng, g), None).reshape(x.shape)
            e = awq_ops.clip_errs_cand(wd, cands, x, xq, g)
            errs = e if errs is None else errs.add_(e)
        if len(inputs) > 1:
            errs /= len(inputs)
        return self._argmin_levels(errs, org_max, org_min, n_grid, levels)

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
