"""Awq with llmc's operator surface (llmc/compression/quantization/awq.py:28-372), arithmetic in HIP.

search_scale_subset keeps the reference's semantics: 20-point ratio grid, fake-quant of the scaled weights in the
model dtype, loss on the INSPECTED module's output, per-batch best bookkeeping (awq.py:229-248), padding
mask, rank-wise winner-takes-all (all_reduce MIN / MAX + broadcast, awq.py:255-273). Two routes, same arithmetic:
  * fused (search_scale_stacked): the inspected module IS the subset's single Linear (o_proj, down_proj in Llama),
    one calibration batch, no kwargs / mask — scale, fake-quant, x/s, GEMM and MSE run as HIP kernels, the 20 losses
    stay on the device;
  * general (inspect_module_forward): any inspected module (whole attention for q/k/v, whole MLP for gate/up,
    llmc/models/llama.py:62,79), any number of batches — scales, weight fake-quant and input scaling are the same
    HIP kernels, the module's own forward runs in torch with the subset's Linear layers routed through the HIP GEMM
    (llmc_linear_eval) for the duration of the search, the loss is formed exactly as calculate_loss does."""
import torch
import torch.distributed as dist
import torch.nn as nn

from llmc_amd.utils.registry_factory import ALGO_REGISTRY

from . import awq_ops
from .awq_pipeline import search_scale_stacked
from .base_blockwise_quantization import BaseBlockwiseQuantization, _world
from .module_utils import (_LLMC_LINEAR_TYPES_, _LLMC_LN_TYPES_, _TRANSFORMERS_LINEAR_TYPES_,
                           _TRANSFORMERS_LN_TYPES_, FakeQuantLinear)
from .quant import IntegerQuantizer


class _hip_linear_forward:
    """While active, `layers` (plain Linear modules of the inspected module) compute y = x W^T (+ b) with the HIP GEMMs
    (awq_ops.linear_auto) instead of the vendor BLAS behind F.linear: the k-tiled one-wave-per-SIMD kernel where the
    shape allows it — the activation tensor q / k / v (gate / up) share is packed k-tiled ONCE per grid point, each
    fake-quantized weight once — else the row-major kernel; shapes neither takes (K % 64 != 0, operands >= 4 GiB) go to
    the framework's GPU linear (logged once)."""

    def __init__(self, layers):
        self.layers = [l for l in layers if isinstance(l, nn.Linear)]
        self.saved = []
        self.xcache = {}

    def _forward(self, layer, x):
        return awq_ops.linear_auto(x, layer.weight.data, None if layer.bias is None else layer.bias.data, self.xcache)

    def __enter__(self):
        for l in self.layers:
            self.saved.append((l, l.__dict__.get('forward')))
            l.forward = (lambda x, _l=l: self._forward(_l, x))
        return self

    def __exit__(self, *exc):
        self.xcache.clear()
        for l, f in self.saved:
            if f is None:
                l.__dict__.pop('forward', None)
            else:
                l.forward = f
        return False


@ALGO_REGISTRY
class Awq(BaseBlockwiseQuantization):
    def __init__(self, model, quant_config, input, padding_mask, config):
        super().__init__(model, quant_config, input, padding_mask, config)
        special = self.quant_config.get('special', {}) or {}
        self.trans = special.get('trans', True)
        self.trans_version = special.get('trans_version', 'v2')
        self.save_scale = special.get('save_scale', False)
        self.awq_bs = special.get('awq_bs', None)
        self.save_mem = special.get('save_mem', True)

    def _fusable_wquantizer(self):
        """llmc_awq_scale_fakequant / search_scale_stacked evaluate integer min/max quantizers with one range per row
        or per group; every other weight quantizer (FP8 e4m3 / e5m2, per_tensor, mse ranges) takes the two-kernel form
        mul_cols + fake_quant_weight_dynamic — same arithmetic as awq.py:155-156, one more HBM pass."""
        wq = self.wquantizer
        return (isinstance(wq, IntegerQuantizer) and wq.granularity in ('per_group', 'per_channel')
                and wq.calib_algo == 'minmax' and wq.round_zp)

    def _fake_quantize_weight(self, w0, cols, s0=None):
        """fake_quantize_weight (awq.py:147-164) of one layer from its original weights; w0 is not modified. A block-wise
        FP8 checkpoint weight (w0 float8_e4m3fn, s0 = its weight_scale_inv) is de-blocked to bf16, scaled, fake-quantized and
        re-blocked: returns (fp8 weight, new weight_scale_inv) then (awq.py:148-161)."""
        if w0.dtype == torch.float8_e4m3fn:
            tmp = self._fp8_to_bf16(w0, s0)
            tmp = self.wquantizer.fake_quant_weight_dynamic(awq_ops.mul_cols_(tmp, cols.to(tmp.dtype)))
            return self._bf16_to_fp8(tmp)
        if self._fusable_wquantizer():
            return awq_ops.scale_fakequant(w0, cols, self.wquantizer)
        return self.wquantizer.fake_quant_weight_dynamic(awq_ops.mul_cols_(w0.clone(), cols))

    @torch.no_grad()
    def scaling_weight(self, w, scales, is_gqa):
        """awq.py:40-46: w *= scales[None, :] in place (per key/value channel repeated per query-head group under GQA)."""
        cols = self.repeat_gqa_scales(scales) if is_gqa else scales
        return awq_ops.mul_cols_(w, cols.reshape(-1).to(w.dtype).contiguous())

    @torch.no_grad()
    def fake_quantize_weight(self, fc, scales, is_gqa, layer_name):
        """awq.py:147-164: fc.weight.data <- fakequant(fc.weight.data * scales) (a block-wise FP8 checkpoint weight is
        de-blocked, scaled, fake-quantized and re-blocked together with weight_scale_inv); returns fc.weight. The tensor
        fc.weight.data pointed at before the call is left untouched (the reference scales it in place and then drops it)."""
        cols = (self.repeat_gqa_scales(scales) if is_gqa else scales).reshape(-1).contiguous()
        if self._is_fp8(fc):
            fc.weight.data, fc.weight_scale_inv.data = self._fake_quantize_weight(fc.weight.data, cols, fc.weight_scale_inv.data)
        else:
            fc.weight.data = self._fake_quantize_weight(fc.weight.data, cols)
        return fc.weight

    def fake_quantize_input(self, x_tmp, layers_dict=None):
        """awq.py:166-177: dynamic activation fake-quant of the scaled input — the whole batch when it is one awq_bs
        batch, else sample by sample (a per_tensor range is then per sample)."""
        if self._bs == x_tmp.shape[0]:
            return self.aquantizer.fake_quant_act_dynamic(x_tmp)
        return torch.stack([self.aquantizer.fake_quant_act_dynamic(x_tmp[i]) for i in range(x_tmp.shape[0])])

    @torch.no_grad()
    def get_weight_scale(self, layers_dict):
        g = self.wquantizer.group_size if self.wquantizer.granularity == 'per_group' else 0
        total = None
        for m in layers_dict.values():
            w = self._fp8_to_bf16(m.weight, m.weight_scale_inv) if self._is_fp8(m) else m.weight.data   # awq.py:53-58
            s = awq_ops.weight_mean(w, g)
            total = s if total is None else total.add_(s)
        return total.div_(len(layers_dict))

    @torch.no_grad()
    def get_act_scale(self, x):
        return awq_ops.act_mean(x)

    @torch.no_grad()
    def get_scales(self, prev_op, x, w_max, is_gqa, ratio):
        if is_gqa:
            # awq.py:89-91, 104-105: the scales of a GQA v_proj -> o_proj subset come from v_proj's OUTPUT (one per key/value
            # channel), always in the v2 form
            x = awq_ops.linear_auto(x, prev_op.weight.data, getattr(prev_op, 'bias', None))
            return awq_ops.awq_scales(self._act_scale_batched(x), None, ratio, 'v2')
        return awq_ops.awq_scales(self.get_act_scale(x), w_max, ratio, self.trans_version)

    # ---- the reference's helper surface (awq.py:110-145), used by the general route ---------------------------
    def inspect_module_forward(self, x, inspect_module, kwargs):
        if self._bs == x.shape[0]:
            out = inspect_module(x, **kwargs)
            return out[0] if isinstance(out, tuple) else out
        outs = []
        for num in range(x.shape[0] // self._bs):
            out = inspect_module(x[num * self._bs:(num + 1) * self._bs], **kwargs)
            outs.append(out[0] if isinstance(out, tuple) else out)
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def get_original_out(self, x, inspect_module, subset_kwargs):
        return self.inspect_module_forward(x, inspect_module, subset_kwargs)

    def calculate_loss(self, org_out, out):
        """mean((org - out)^2) in fp32, averaged over sub-batches of awq_bs (awq.py:134-145); a 0-dim device tensor
        (the reference calls .item() here: 20 x batches host syncs; the comparison below needs one per grid point)."""
        if out.shape[0] == self._bs:
            return (org_out - out).float().pow(2).mean()
        total, b_num = 0.0, org_out.shape[0] // self._bs
        for num in range(b_num):
            sl = slice(num * self._bs, (num + 1) * self._bs)
            total = total + (org_out[sl] - out[sl]).float().pow(2).mean()
        return total / b_num

    def _act_scale_batched(self, x):
        """get_act_scale (awq.py:74-85): mean of sub-batch means when awq_bs splits the batch."""
        if x.shape[0] == self._bs:
            return awq_ops.act_mean(x)
        means = [awq_ops.act_mean(x[n * self._bs:(n + 1) * self._bs]) for n in range(x.shape[0] // self._bs)]
        return sum(means) / len(means)

    def _fused_route_ok(self, layers_dict, input, inspect_module, subset_kwargs):
        layers = list(layers_dict.values())
        if len(input) != 1 or len(layers) != 1 or inspect_module is not layers[0]:
            return False
        if not self.w_only or not self._fusable_wquantizer():
            return False
        if self.padding_mask or (isinstance(subset_kwargs, dict) and subset_kwargs) or isinstance(subset_kwargs, list):
            return False
        if getattr(layers[0], 'bias', None) is not None:
            return False       # a bias cancels in org - out, but the fused loss kernel takes the bias-free product
        if any(self._is_fp8(l) for l in layers):
            return False       # block-wise FP8 checkpoints: the layer's own fp8 forward evaluates the re-blocked weight
        return self.awq_bs is None or self.awq_bs == input[0].shape[0]

    @torch.no_grad()
    def search_scale_subset(self, prev_op, layers_dict, input, inspect_module, is_gqa, subset_kwargs):
        self._bs = input[0].shape[0] if self.awq_bs is None else self.awq_bs
        if is_gqa:
            best_scales, best_error = self._search_scale_general(layers_dict, input, inspect_module, subset_kwargs,
                                                                 prev_op=prev_op, is_gqa=True)
        elif self._fused_route_ok(layers_dict, input, inspect_module, subset_kwargs):
            x = input[0]
            weights = [fc.weight.data for fc in layers_dict.values()]
            best_scales, losses, n = search_scale_stacked(weights, x, self.wquantizer, self.trans_version,
                                                          return_losses=True)
            best_error = losses[n].reshape(1).clone()
        else:
            best_scales, best_error = self._search_scale_general(layers_dict, input, inspect_module, subset_kwargs)
        if _world() > 1:   # winner-takes-all across ranks (awq.py:255-273)
            gbest = best_error.clone()
            dist.all_reduce(gbest, op=dist.ReduceOp.MIN)
            rank = torch.tensor([dist.get_rank() if abs(float(best_error) - float(gbest)) < 1e-5 else -1],
                                device=best_scales.device)
            dist.all_reduce(rank, op=dist.ReduceOp.MAX)
            best_scales = best_scales.clone()
            dist.broadcast(best_scales, src=int(rank.item()))
        return best_scales

    @torch.no_grad()
    def _search_scale_general(self, layers_dict, input, inspect_module, subset_kwargs, n_grid=20, prev_op=None, is_gqa=False):
        """awq.py:189-253 with the module kept on the device: weights are restored from a device copy after every
        evaluation (the reference reloads a CPU state dict), everything else in the reference's order. is_gqa (do_gqa_trans):
        the scales are per key/value channel of `prev_op` (v_proj) and reach the layer and its input repeated per query-head
        group (repeat_gqa_scales)."""
        layers = list(layers_dict.values())
        w_max = self.get_weight_scale(layers_dict)
        org_w = [fc.weight.data.clone() for fc in layers]
        org_s = [fc.weight_scale_inv.data.clone() if self._is_fp8(fc) else None for fc in layers]
        best_error, best_scales = float('inf'), None
        org_out_dict = {}
        dev = org_w[0].device
        with _hip_linear_forward(layers):
            for n in range(n_grid):
                loss_mean, scales_mean = 0, 0
                for i in range(len(input)):
                    x = input[i] = input[i].to(dev)
                    kwargs = subset_kwargs[i] if isinstance(subset_kwargs, list) else (subset_kwargs or {})
                    if i not in org_out_dict:
                        org_out_dict[i] = self.get_original_out(x, inspect_module, kwargs)
                    org_out = org_out_dict[i]
                    ratio = n * 1 / n_grid
                    if is_gqa:
                        scales = self.get_scales(prev_op, x, w_max, True, ratio)
                        cols = self.repeat_gqa_scales(scales).reshape(-1).contiguous()
                    else:
                        scales = awq_ops.awq_scales(self._act_scale_batched(x), w_max, ratio, self.trans_version)
                        cols = scales
                    for lname, fc in layers_dict.items():             # awq.py:218-219; fc.weight.data is the original here
                        self.fake_quantize_weight(fc, scales, is_gqa, lname)
                    x_tmp = awq_ops.div_cols(x, cols)     # scaling_input (base_blockwise_quantization.py:877-889)
                    if not self.w_only:
                        x_tmp = self.fake_quantize_input(x_tmp, layers_dict)      # awq.py:223-224
                    out = self.inspect_module_forward(x_tmp, inspect_module, kwargs)
                    if self.padding_mask and org_out.shape[1] == self.padding_mask[i].shape[-1]:
                        m = self.padding_mask[i].unsqueeze(dim=-1).to(org_out.device)
                        org_out, out = org_out * m, out * m
                    loss = self.calculate_loss(org_out, out)
                    for fc, w0, s0 in zip(layers, org_w, org_s):      # inspect_module.load_state_dict(org_sd)
                        fc.weight.data = w0
                        if s0 is not None:
                            fc.weight_scale_inv.data = s0
                    if len(input) == 1:
                        # one batch (the shipped bs: -1): loss_mean = loss, scales_mean = scales; the 20 losses and
                        # the running best stay on the device — no host sync per grid point (the reference's .item())
                        # Like the reference (best_error = inf, `loss_mean < best_error`, awq.py:189,245): a NaN loss never
                        # wins, whichever grid point it occurs at, and the first finite minimum is kept.
                        loss = loss.reshape(1).float()
                        if best_scales is None:
                            best_err_dev = torch.full_like(loss, float('inf'))
                            best_scales = torch.zeros_like(scales)
                        better = loss < best_err_dev
                        best_scales = torch.where(better, scales, best_scales)
                        best_err_dev = torch.where(better, loss, best_err_dev)
                        continue
                    loss = float(loss)
                    n_samples = self.n_samples
                    loss_mean += x.shape[0] * 1.0 / n_samples * loss
                    scales_mean += x.shape[0] * 1.0 / n_samples * scales   # in place from the 2nd batch on: `best_scales` below aliases it, like the reference
                    if loss_mean < best_error:             # inside the batch loop, like the reference
                        best_error, best_scales = loss_mean, scales_mean
        if len(input) == 1:
            if not bool(torch.isfinite(best_err_dev).all()):      # the subset's one host sync; the reference ends with best_scales = None here
                raise RuntimeError('AWQ scale search: no grid point produced a finite loss')
            return best_scales, best_err_dev
        return best_scales, torch.tensor([best_error], dtype=torch.float32, device=dev)

    @torch.no_grad()
    def block_transform(self, block, input_feat, block_kwargs):
        if self.trans:
            super().block_transform(block, input_feat, block_kwargs)
        if self.weight_clip:
            self.auto_clipper.run(block, self.block_idx, input_feat,
                                  n_sample_token=self.config.calib.get('seq_len', None))

    @torch.no_grad()
    def subset_transform(self, subset, input_feat, subset_kwargs):
        layers_dict = subset['layers']
        prev_op = subset['prev_op']
        input_name = subset['input'][0]
        if not subset.get('do_trans', True):
            return
        assert len(prev_op) in (0, 1), 'Only support single prev_op. If multi prev_ops, code need to be updated.'
        if len(prev_op) == 0 or prev_op[0] is None:
            return
        ln_types = tuple(_LLMC_LN_TYPES_ + _TRANSFORMERS_LN_TYPES_)
        lin_types = tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_)
        if not isinstance(prev_op[0], ln_types + lin_types) and not hasattr(prev_op[0], 'weight'):
            return
        layers = list(layers_dict.values())
        is_gqa = False
        if isinstance(prev_op[0], (nn.Linear, FakeQuantLinear)):
            of, inf = prev_op[0].out_features, layers[0].in_features
            if of not in (inf * 3, inf * 2, inf):
                if getattr(self, 'has_gqa', False) and getattr(self, 'do_gqa_trans', False):
                    # awq.py:338-343: v_proj -> o_proj of a GQA attention; the search runs on the inputs of the previous subset
                    is_gqa = True
                    input_keys = list(input_feat.keys())
                    input_name = input_keys[input_keys.index(input_name) - 1]
                else:
                    return                                                  # awq.py:344-346: "Cannot apply scale"

        scale = self.search_scale_subset(prev_op[0], layers_dict, input_feat[input_name], subset['inspect'], is_gqa,
                                         subset_kwargs)
        self.apply_scale(scale, prev_op, layers)
        self.update_input_feat(scale, input_feat, layers_dict, is_gqa)
        if self.save_scale:
            for n in layers_dict:
                self.act_scales[f'{self.model.block_name_prefix}.{self.block_idx}.{n}'] = scale
