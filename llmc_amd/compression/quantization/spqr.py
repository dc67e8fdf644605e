"""SpQR with llmc's operator surface (llmc/compression/quantization/spqr.py:18-398), arithmetic in HIP.

Same: constructor, `quant.special` keys (actorder, percdamp, blocksize, true_sequential, relative_threshold,
simplified_outliers, scale, zero), hook protocol, module buffers buf_scales / buf_zeros / buf_qmax / buf_qmin /
buf_perm / buf_invperm / buf_mask (sparse fp32, spqr.py:181), fp32 `layer.weight.data` after the transform, w_qdq,
deploy refusing real_quant. Different in mechanics only: the Hessian machinery is GPTQ's (one accumulator per distinct
input tensor, K1 on the MFMA pipe, one factorisation per shared input instead of one per layer), the column loop is
`llmc_spqr_quantize` (spqr_loop.hip) on the stacked rows of the layers that share the factor.

As in the reference, only asymmetric per-group weights work (its get_group_qparams hands the symmetric quantizer's
0-dim zero point to `reshape_tensor`, which raises), and the second-level scale / zero quantizers act on [R, 1]
tensors, i.e. never group anything (spqr.py:323-345): `llmc_spqr_quantize` reproduces what they do return.
"""
import copy
import math
from dataclasses import dataclass

import torch

from llmc_amd import _ffi
from llmc_amd.utils.registry_factory import ALGO_REGISTRY

from . import gptq_ops
from .base_blockwise_quantization import BaseBlockwiseQuantization
from .gptq import GPTQ
from .quant import IntegerQuantizer


@dataclass
class SpqrConfig:
    bit: int = 4
    group_size: int = 16
    actorder: bool = True
    percdamp: float = 1.0
    blocksize: int = 128
    relative_threshold: float = 0.2        # math.inf: no outliers
    simplified_outliers: bool = False
    scale_bit: int = 3
    zero_bit: int = 3


@dataclass
class SpqrResult:
    weight: torch.Tensor      # [R, K] fp32 tmp, original column order
    mask: torch.Tensor        # [R, K] bool, original column order
    scales: torch.Tensor      # [R, K/g] fp32, processing order
    zeros: torch.Tensor       # [R, K/g] fp32
    perm: torch.Tensor        # [K] | None
    loss: torch.Tensor        # 0-dim fp32: sum(Losses) (spqr.py:166)
    threshold: float
    info: torch.Tensor


def spqr_quantize(W, U, cfg, threshold):
    """spqr.py:185-254. W [R, K] fp32 (overwritten: running weights), U [K, K] fp32 upper factor.
    Returns (tmp, losses, mask uint8, scales [R, K/g], zeros)."""
    _ffi.require_gpu(W, U)
    L = _ffi.lib()
    R, K = W.shape
    dev = W.device
    ng = K // cfg.group_size
    tmp, losses = torch.empty_like(W), torch.empty_like(W)
    mask = torch.empty((R, K), dtype=torch.uint8, device=dev)
    scales = torch.empty((R, ng), dtype=torch.float32, device=dev)
    zeros = torch.empty((R, ng), dtype=torch.float32, device=dev)
    ws = _ffi.workspace(L.llmc_spqr_quantize_ws_bytes(R, K), dev)
    _ffi.check(L.llmc_spqr_quantize(
        _ffi.ptr(W), _ffi.ptr(U), R, K, 0.0, float(2 ** cfg.bit - 1), int(cfg.group_size), float(threshold),
        int(bool(cfg.simplified_outliers)), 0.0, float(2 ** cfg.scale_bit - 1), 0.0, float(2 ** cfg.zero_bit - 1),
        _ffi.ptr(scales), _ffi.ptr(zeros), _ffi.ptr(tmp), _ffi.ptr(losses), _ffi.ptr(mask), int(cfg.blocksize),
        _ffi.ptr(ws), _ffi.stream()), 'llmc_spqr_quantize')
    return tmp, losses, mask, scales, zeros


def spqr_factor(H, Wcat, cfg):
    """spqr.py:129-161: actorder permutation, damping by percdamp * mean(|diag|) with the dead zeros still in it, THEN
    dead diagonal := 1 and dead weight columns := 0, chol -> inverse -> chol(upper). Returns (perm | None, Wp, U, info)."""
    d0 = torch.diagonal(H)
    perm = torch.argsort(d0, descending=True) if cfg.actorder else None
    dp = d0 if perm is None else d0[perm]
    dead = dp == 0
    mean_abs = dp.abs().mean()
    # gather + dead fix on the kernel GPTQ uses (no damping there: SpQR damps before it fixes the dead diagonal)
    Hp, Wp = gptq_ops.hessian_prep(H, Wcat, perm, 0.0, want_h=True)
    dg = torch.diagonal(Hp)
    if cfg.percdamp > 0:
        dg += cfg.percdamp * mean_abs
    dg[dead] = 1.0
    U, info = gptq_ops.chol_inv_upper(Hp, check=False, return_info=True)
    return perm, Wp, U, info


def quantize_stacked(W_list, H, cfg):
    """The layers of W_list share the input whose Hessian is H: one factorisation; the outlier threshold is per layer
    (spqr.py:205-206 uses the layer's own W), so the column loop runs per layer on the shared factor."""
    _ffi.require_gpu(H, *W_list)
    Wcat = torch.cat([w.reshape(w.shape[0], -1) for w in W_list], dim=0) if len(W_list) > 1 else W_list[0]
    perm, Wp, U, info = spqr_factor(H, Wcat, cfg)
    invperm = torch.argsort(perm) if perm is not None else None
    du2 = torch.diagonal(U).square()
    out, r0 = [], 0
    for w in W_list:
        R = w.shape[0]
        Wl = Wp[r0:r0 + R].contiguous()
        r0 += R
        thr = math.inf
        if cfg.relative_threshold != math.inf:
            # spqr.py:205-206 (`.item()`: one host sync per layer, like the reference)
            thr = cfg.relative_threshold * (Wl.var(dim=0) / du2).mean().item()
        tmp, losses, mask, s, z = spqr_quantize(Wl, U, cfg, thr)
        if invperm is not None:
            K = tmp.shape[1]
            tmp = gptq_ops.gather_cols(tmp, invperm) if (K % 4 == 0 and K <= gptq_ops.GATHER_MAX_K) else tmp.index_select(1, invperm)
            mask = mask.index_select(1, invperm)
        out.append(SpqrResult(weight=tmp, mask=mask.bool(), scales=s, zeros=z, perm=perm, loss=losses.sum(),
                              threshold=thr, info=info))
    return out


@ALGO_REGISTRY
class SpQR(GPTQ):
    def __init__(self, model, quant_config, input, padding_mask, config, modality='language'):
        BaseBlockwiseQuantization.__init__(self, model, quant_config, input, padding_mask, config)
        assert self.wquantizer.granularity == 'per_group', 'SpQR only supports per_group quantization'
        self.model_dtype = next(self.model.model.parameters()).dtype
        self.add_quant_config()
        self.layers_cache = {}
        self._groups, self._group_of = {}, {}

    @torch.no_grad()
    def add_quant_config(self):
        special = self.quant_config['special']
        self.true_sequential = special['true_sequential']
        self.actorder = special['actorder']
        self.percdamp = special['percdamp']
        self.blocksize = special['blocksize']
        self.relative_threshold = special['relative_threshold']
        self.simplified_outliers = special['simplified_outliers']
        self.static_groups = False
        self.owq = False
        self.chunk_num = 1
        if self.actorder:
            self.need_perm = True
        else:
            self.need_perm = False
        if self.relative_threshold == 'inf':
            self.relative_threshold = math.inf
        self.quant_type = self.quant_config.get('quant_type', 'int-quant')
        assert self.quant_type != 'float-quant', 'SPQR do not support Float quant now.'
        if self.wquantizer.sym:
            raise NotImplementedError('SpQR needs an asymmetric weight quantizer (the reference crashes on the symmetric '
                                      'quantizer\'s 0-dim zero point in get_group_qparams, spqr.py:331)')
        if getattr(self.wquantizer, 'round_zp', True):
            # the column loop's kernel evaluates clamp(round(x / s + z)) with a fractional zero point (the shipped
            # configs: round_zp False); with round_zp the reference rounds and clamps z and uses clamp(round(x / s) + z)
            raise NotImplementedError('SpQR weight quantizer: only round_zp=False (the shipped configs) is built')
        self.scale_quantizer = IntegerQuantizer(**special['scale'])
        self.zero_quantizer = IntegerQuantizer(**special['zero'])
        for q, what in ((self.scale_quantizer, 'scale'), (self.zero_quantizer, 'zero')):
            if q.sym or q.granularity == 'per_tensor' or getattr(q, 'round_zp', True):
                raise NotImplementedError(f'SpQR {what} quantizer: only the asymmetric, round_zp=False, per_group / '
                                          'per_channel form of the shipped configs is built')
        self.scfg = SpqrConfig(bit=self.wquantizer.bit, group_size=self.wquantizer.group_size, actorder=self.actorder,
                               percdamp=self.percdamp, blocksize=self.blocksize,
                               relative_threshold=self.relative_threshold, simplified_outliers=self.simplified_outliers,
                               scale_bit=self.scale_quantizer.bit, zero_bit=self.zero_quantizer.bit)

    def _transform_group(self, gid, layers, names):
        H = self._groups[gid]['acc'].H
        results = quantize_stacked([l.weight.data for l in layers], H, self.scfg)
        gptq_ops.raise_if_not_pd(results[0].info, 'SpQR: Hessian of ' + ', '.join(names))
        self.last_losses, self.last_outliers = {}, {}
        for l, n, r in zip(layers, names, results):
            l.weight.data = r.weight.reshape(l.weight.shape)                  # fp32 tmp (spqr.py:174)
            self.last_losses[n] = r.loss
            self.last_outliers[n] = r.mask.sum()
            if self.actorder:
                l.register_buffer('buf_perm', r.perm)
                l.register_buffer('buf_invperm', torch.argsort(r.perm))
            l.register_buffer('buf_scales', r.scales.reshape(-1, 1).clone())   # set_model_qparams (spqr.py:347-355)
            l.register_buffer('buf_zeros', r.zeros.reshape(-1, 1).clone())
            l.register_buffer('buf_qmax', torch.tensor(float(self.wquantizer.qmax)))
            l.register_buffer('buf_qmin', torch.tensor(float(self.wquantizer.qmin)))
            l.register_buffer('buf_mask', r.mask.float().to_sparse())

    # ---- the reference's per-layer method surface (spqr.py:116-254, 315-355): same names, arguments and in-place effects ----
    _REFERENCE_HOOKS = ('weight_transform', 'get_group_qparams', 'set_model_qparams', 'merge_qparams')

    @torch.no_grad()
    def weight_transform(self, W, Hinv, Losses, tmp, mask):
        """spqr.py:185-254 (llmc_spqr_quantize): in place like the reference — `tmp` the compensated weights, `Losses` the
        squared errors, `mask` the outliers, `W` the running weights; `self.groups[g]` / `self.qparams` hold every group's
        second-level-quantized scales / zeros (get_group_qparams)."""
        thr = math.inf
        if self.relative_threshold != math.inf:
            thr = self.relative_threshold * (W.var(dim=0) / torch.diag(Hinv).square()).mean().item()     # spqr.py:205-206
        if not W.is_contiguous():
            raise ValueError('SpQR.weight_transform: W must be contiguous (it is updated in place)')
        t, l, m, s, z = spqr_quantize(W, Hinv.contiguous(), self.scfg, thr)
        tmp.copy_(t)
        Losses.copy_(l)
        mask.copy_(m.to(mask.dtype))
        qmax, qmin = self.wquantizer.qmax.to(W.device), self.wquantizer.qmin.to(W.device)
        self.groups = [{'scales': s[:, g:g + 1].clone(), 'zeros': z[:, g:g + 1].clone(), 'qmax': qmax, 'qmin': qmin}
                       for g in range(s.shape[1])]
        self.qparams = dict(self.groups[-1])
        self.last_threshold = thr

    @torch.no_grad()
    def merge_qparams(self, qparams):
        if isinstance(qparams, int):
            return qparams
        if self.wquantizer.granularity == 'per_group':
            qparams = torch.stack(qparams, dim=1).reshape(-1, 1)
        return qparams

    @torch.no_grad()
    def get_group_qparams(self, c_tensor, idx):
        """spqr.py:323-345: min/max qparams of one column group, then the second-level scale / zero quantizers (which see
        [R, 1] tensors, i.e. one value per row: their fake-quant returns what the reference's returns). HIP quantizer
        kernels; kept in `self.qparams` and `self.groups[idx // group_size]`."""
        _, s, z, qmax, qmin = self.wquantizer.get_tensor_qparams(c_tensor.contiguous())
        _, ss, zs, Ps, Ns = self.scale_quantizer.get_tensor_qparams(s)
        scales = self.scale_quantizer.fake_quant_weight_static(s.data, {'scales': ss, 'zeros': zs, 'qmin': Ns, 'qmax': Ps})
        _, sz, zz, Pz, Nz = self.zero_quantizer.get_tensor_qparams(z)
        zeros = self.zero_quantizer.fake_quant_weight_static(z.data, {'scales': sz, 'zeros': zz, 'qmin': Nz, 'qmax': Pz})
        self.qparams = {'scales': scales, 'zeros': zeros, 'qmax': qmax, 'qmin': qmin}
        if not isinstance(getattr(self, 'groups', None), list):
            self.groups = [None] * (self.columns // self.wquantizer.group_size)
        self.groups[idx // self.wquantizer.group_size] = copy.deepcopy(self.qparams)

    @torch.no_grad()
    def set_model_qparams(self, layer):
        """spqr.py:347-355."""
        layer.register_buffer('buf_scales', copy.deepcopy(self.merge_qparams([g['scales'] for g in self.groups])))
        layer.register_buffer('buf_zeros', copy.deepcopy(self.merge_qparams([g['zeros'] for g in self.groups])))
        layer.register_buffer('buf_qmax', torch.tensor(float(self.groups[0]['qmax'])))
        layer.register_buffer('buf_qmin', torch.tensor(float(self.groups[0]['qmin'])))

    @torch.no_grad()
    def layer_transform_reference(self, layer, name):
        """spqr.py:116-183: the reference's per-layer flow through weight_transform / set_model_qparams."""
        self.qparams = {}
        self.columns = self.layers_cache[name]['columns']
        W = layer.weight.data
        if W.dim() > 2:
            W = W.flatten(1)
        self.groups = [None] * (self.columns // self.wquantizer.group_size)
        perm, Wp, U, info = spqr_factor(self._hessian_of(name), W.contiguous(), self.scfg)
        gptq_ops.raise_if_not_pd(info, f'SpQR: Hessian of {name}')
        if perm is not None:
            self.perm, self.invperm = perm, torch.argsort(perm)
            layer.register_buffer('buf_perm', self.perm)
            layer.register_buffer('buf_invperm', self.invperm)
        Losses, tmp = torch.zeros_like(Wp), torch.zeros_like(Wp)
        mask = torch.zeros_like(Wp, dtype=torch.bool)
        self.weight_transform(Wp, U, Losses, tmp, mask)
        self.last_losses = getattr(self, 'last_losses', {})
        self.last_losses[name] = Losses.sum()
        if perm is not None:
            tmp, mask = tmp[:, self.invperm], mask[:, self.invperm]
        layer.weight.data = tmp.reshape(layer.weight.shape)
        self.set_model_qparams(layer)
        layer.register_buffer('buf_mask', mask.float().to_sparse())

    @torch.no_grad()
    def block_transform_true_sequential(self, block, input_feat):
        """spqr.py:61-91: subset by subset, re-hooking the block after each (the base class's block loop with
        true_sequential on: rehook_next_subset)."""
        saved, self.true_sequential = self.true_sequential, True
        try:
            BaseBlockwiseQuantization.block_transform(self, block, input_feat, {})
        finally:
            self.true_sequential = saved

    @torch.no_grad()
    def collect_model_qparams(self):
        pass

    @torch.no_grad()
    def w_q(self, module, wquantizer):
        pass

    @torch.no_grad()
    def w_qdq(self, module, wquantizer):
        """spqr.py:357-380: outliers keep their (compensated) weight, the rest is fake-quantized with the stored qparams."""
        mask = module.buf_mask.to_dense()
        weight = module.weight
        out = (mask * weight).to(self.model_dtype)
        if self.need_perm:
            weight = weight[:, module.buf_perm]
        args = {'scales': module.buf_scales, 'zeros': module.buf_zeros, 'qmax': module.buf_qmax, 'qmin': module.buf_qmin}
        weight = wquantizer.fake_quant_weight_static(weight, args).to(self.model_dtype)
        if self.need_perm:
            weight = weight[:, module.buf_invperm]
        return (weight * (1 - mask) + out).to(self.model_dtype)

    @torch.no_grad()
    def deploy(self, quant_format):
        if quant_format == 'real_quant':
            assert False, 'SpQR does not support real quantization'
        BaseBlockwiseQuantization.deploy(self, quant_format)
