"""Quantizers with llmc's operator surface, arithmetic in HIP (libllmc_hip.so).

Mirrors llmc/compression/quantization/quant.py: class names, constructor kwargs, method names,
argument meaning, return shapes/dtypes (BaseQuantizer :46-658, IntegerQuantizer :661-960,
FloatQuantizer :963-1229).  In scope: calib_algo 'minmax' (the algorithm of every GPTQ/AWQ/RTN config
named in BASELINE.json); granularity per_group / per_channel / per_token / per_tensor / per_head;
FloatQuantizer e4m3 / e5m2 with qtorch.float_quantize restated (fp8_semantics='cast': torch's dtype cast).
Also in scope: calib_algo 'mse' (get_mse_range, quant.py:145-203).
Also in scope: calib_algo 'learnable' (get_learnable_range, quant.py:205-224: the range AutoClipper's clip v2 factors scale).
Out of scope (raise NotImplementedError): hqq range search, W48.

Tensors must live on the GPU; there is no CPU fallback (see llmc_amd/_ffi.py).
"""
import torch

from llmc_amd import _ffi


def _rows(t):
    """[G, g] geometry of an already reshaped tensor (reduction over the last dim)."""
    g = t.shape[-1]
    return t.numel() // g, g


class BaseQuantizer(object):
    def __init__(self, bit, symmetric, granularity, **kwargs):
        self.bit = bit
        self.sym = symmetric
        self.granularity = granularity
        self.kwargs = kwargs

        self.calib_algo = self.kwargs.get('calib_algo', 'minmax')
        if self.calib_algo not in ('minmax', 'static_minmax', 'static_moving_minmax', 'static_hist', 'mse', 'learnable'):
            raise NotImplementedError(f'calib_algo={self.calib_algo}: hqq ranges are outside the hot path')
        # hist config (quant.py:81-86)
        self.bins = self.kwargs.get('bins', 2048)
        self.upsample_rate = self.kwargs.get('upsample_rate', 16)
        # mse config (quant.py:78-81)
        self.mse_b_num = self.kwargs.get('mse_b_num', 1)
        self.maxshrink = self.kwargs.get('maxshrink', 0.8)
        self.mse_grid = self.kwargs.get('mse_grid', 100)

        if self.granularity == 'per_group':
            self.group_size = self.kwargs['group_size']
        elif self.granularity == 'per_head':
            self.head_num = self.kwargs['head_num']
        elif self.granularity == 'per_block':
            self.block_size = self.kwargs['block_size']

        if self.kwargs.get('ste', False) or self.kwargs.get('ste_all', False):
            raise NotImplementedError('STE rounding (training-time) is outside the hot path')
        self.round_zp = self.kwargs.get('round_zp', True)

    # -- views (quant.py:612-658) ---------------------------------------------------------------
    def reshape_tensor(self, tensor, allow_padding=False):
        if self.granularity == 'per_group':
            if tensor.shape[-1] >= self.group_size:
                if tensor.shape[-1] % self.group_size == 0:
                    t = tensor.reshape(-1, self.group_size)
                elif allow_padding:
                    deficiency = self.group_size - tensor.shape[1] % self.group_size
                    prefix = tensor.shape[:-1]
                    pad_zeros = torch.zeros((*prefix, deficiency), device=tensor.device, dtype=tensor.dtype)
                    t = torch.cat((tensor, pad_zeros), dim=-1).reshape(-1, self.group_size)
                else:
                    raise ValueError(
                        f'Dimension {tensor.shape[-1]} not divisible by group size {self.group_size}')
            else:
                t = tensor
        elif self.granularity == 'per_head':
            t = tensor.reshape(self.head_num, -1)
        else:
            t = tensor
        return t

    def restore_tensor(self, tensor, shape):
        if tensor.shape == shape:
            return tensor
        try:
            return tensor.reshape(shape)
        except RuntimeError:
            deficiency = self.group_size - shape[1] % self.group_size
            return tensor.reshape(*shape[:-1], -1)[..., :-deficiency]

    # -- ranges -> qparams (quant.py:132-143, 545-559), fused in one kernel ------------------------
    def _geometry(self, tensor):
        """(G, g) of the [G, g] reduction view the reference reduces over."""
        if self.granularity == 'per_tensor':
            return 1, tensor.numel()
        return _rows(tensor)

    def _qparam_shape(self, tensor):
        if self.granularity == 'per_tensor':
            return ()
        return (*tensor.shape[:-1], 1)

    def _mse_qparams(self, tensor, want_range=False, norm=2.4):
        """get_mse_range + get_qparams (quant.py:145-203, 545-559): fp32 scales / zeros whatever the tensor dtype
        (the reference searches on tensor.float()). mse_b_num only batches the reference's memory use."""
        _ffi.require_gpu(tensor)
        L = _ffi.lib()
        tensor = tensor.contiguous()
        G, g = self._geometry(tensor)
        if self.mse_b_num < 1 or tensor.shape[0] % self.mse_b_num != 0:
            raise AssertionError('Batch number must be divisible by tensor.shape[0],')
        dev = tensor.device
        scales = torch.empty(G, dtype=torch.float32, device=dev)
        zeros = None if self.sym else torch.empty(G, dtype=torch.float32, device=dev)
        mn = torch.empty(G, dtype=torch.float32, device=dev) if want_range else None
        mx = torch.empty(G, dtype=torch.float32, device=dev) if want_range else None
        _ffi.check(L.llmc_mse_qparams(
            _ffi.ptr(tensor), _ffi.dt(tensor), G, g, int(self.sym), int(self.round_zp), float(self.qmin),
            float(self.qmax), int(self.maxshrink * self.mse_grid), int(self.mse_grid), float(norm), _ffi.ptr(scales),
            _ffi.ptr(zeros), _ffi.ptr(mn), _ffi.ptr(mx), _ffi.stream()), 'llmc_mse_qparams')
        shp = self._qparam_shape(tensor)
        scales = scales.reshape(shp)
        zeros = torch.tensor(0.0) if self.sym else zeros.reshape(shp)
        if want_range:
            return scales, zeros, mn.reshape(shp), mx.reshape(shp)
        return scales, zeros

    # ---- calib_algo 'learnable' (quant.py:205-224, 545-559): the range is min / max scaled by sigmoid(bound factors) — the
    # factors AutoClipper's clip_version v2 stores as buf_upbound_factor / buf_lowbound_factor (auto_clip.py:213-256). With
    # no factors it is the plain min/max range. Products are formed in fp32 and then cast (ATen's CPU kernels round twice;
    # a fused fp16 multiply on the GPU rounds once: csrc/common.h).
    @staticmethod
    def _mul16(a, b):
        return (a.float() * b.float()).to(torch.promote_types(a.dtype, b.dtype))

    def get_learnable_range(self, tensor, lowbound_factor=None, upbound_factor=None):
        min_val, max_val = tensor.amin(dim=-1, keepdim=True), tensor.amax(dim=-1, keepdim=True)
        if self.granularity == 'per_tensor':
            min_val, max_val = torch.min(tensor), torch.max(tensor)
        if self.sym:
            if upbound_factor is not None:
                abs_max = torch.max(max_val.abs(), min_val.abs()).clamp(min=1e-5)
                abs_max = self._mul16(torch.sigmoid(upbound_factor), abs_max)
                min_val, max_val = -abs_max, abs_max
        elif upbound_factor is not None and lowbound_factor is not None:
            min_val = self._mul16(torch.sigmoid(lowbound_factor), min_val)
            max_val = self._mul16(torch.sigmoid(upbound_factor), max_val)
        return min_val, max_val

    def get_qparams(self, tensor_range, device):
        """quant.py:545-559 on a given (min, max) range."""
        min_val, max_val = tensor_range
        qmin, qmax = self.qmin.to(device), self.qmax.to(device)
        if self.sym:
            abs_max = torch.max(max_val.abs(), min_val.abs()).clamp(min=1e-5)
            scales = abs_max / qmax
            zeros = torch.tensor(0.0)
        else:
            scales = (max_val - min_val).clamp(min=1e-5) / (qmax - qmin)
            zeros = (qmin - torch.round(min_val / scales)).clamp(qmin, qmax)
            if not self.round_zp:
                zeros = qmin - (min_val / scales)
        return scales, zeros, qmax, qmin

    def get_minmax_range(self, tensor):
        """quant.py:132-143: (min_val, max_val) of the reshaped tensor, one pair per row (0-dim for per_tensor)."""
        if self.granularity == 'per_tensor':
            return torch.min(tensor), torch.max(tensor)
        return tensor.amin(dim=-1, keepdim=True), tensor.amax(dim=-1, keepdim=True)

    def get_mse_range(self, tensor, norm=2.4, bs=256):
        """quant.py:145-203: the shrunk (min, max) with the smallest |fakequant(x) - x|^norm per row (k_mse_qparams; fp32, the
        reference searches on tensor.float()). `bs` only batches the reference's memory use."""
        _, _, mn, mx = self._mse_qparams(tensor, want_range=True, norm=norm)
        return mn, mx

    def get_tensor_range(self, tensor, args={}):
        """quant.py:122-130 for the algorithms on the accelerated path: (min_val, max_val)."""
        if self.calib_algo == 'learnable':
            return self.get_learnable_range(tensor, **{k: v for k, v in args.items() if k in ('lowbound_factor', 'upbound_factor')})
        if self.calib_algo == 'mse':
            return self.get_mse_range(tensor)
        return self.get_minmax_range(tensor)

    # ---- static activation ranges over a list of calibration samples (quant.py:103-120, 221-263, 462-543, 561-586) ----------
    def reshape_batch_tensors(self, act_tensors):
        """quant.py:103-120 -> a list (one entry per module input) of lists of per-sample tensors."""
        assert len(act_tensors) > 0, (
            'Calibration data is insufficient. Please provide more data to ensure '
            'all experts in the MOE receive an adequate number of tokens.')
        if isinstance(act_tensors[0], tuple):
            return [torch.stack(tl) for tl in zip(*act_tensors)]
        if len(act_tensors) == 1:
            return [[act_tensors[0][i] for i in range(act_tensors[0].size(0))]]
        return [list(act_tensors)]

    def get_minmax_stats(self, act_tensors):
        """quant.py:221-251: per module input the fp32 vectors of per-sample min / max — one launch pair over all the
        separately allocated samples (llmc_minmax_samples) instead of one reduction pair per sample."""
        from .hist_range import sample_minmax
        stats = {}
        for idx, tensors in enumerate(act_tensors):
            mn, mx = sample_minmax(list(tensors))
            stats[idx] = {'min': mn, 'max': mx}
        return stats

    def get_static_minmax_range(self, act_tensors):
        """quant.py:253-263: mean over the samples of the per-sample min / max (fp32)."""
        stats = self.get_minmax_stats(self.reshape_batch_tensors(act_tensors))
        return [r['min'].mean() for r in stats.values()], [r['max'].mean() for r in stats.values()]

    def get_static_moving_minmax_range(self, act_tensors, alpha):
        """quant.py:524-543: exponential moving average of the per-sample ranges in the sample dtype. The per-sample min / max
        come from one kernel pass; the recurrence runs on the host on 0-dim tensors of the sample dtype — the arithmetic the
        reference performs on its scalars."""
        from .hist_range import sample_minmax
        mins, maxs = [], []
        for tensors in self.reshape_batch_tensors(act_tensors):
            tensors = list(tensors)
            smn, smx = sample_minmax(tensors)
            smn, smx = smn.cpu().to(tensors[0].dtype), smx.cpu().to(tensors[0].dtype)
            mn = mx = None
            for a, b in zip(smn, smx):
                mn, mx = (a, b) if mn is None else (mn + alpha * (a - mn), mx + alpha * (b - mx))
            mins.append(mn.to(tensors[0].device))
            maxs.append(mx.to(tensors[0].device))
        return mins, maxs

    def get_static_hist_range(self, act_tensors):
        """quant.py:462-522: histogram-observer range (data passes in HIP: hist_range.static_hist_range)."""
        from .hist_range import static_hist_range
        mins, maxs = [], []
        for tensors in self.reshape_batch_tensors(act_tensors):
            tensors = list(tensors)
            lo, hi = static_hist_range(tensors, self.bins, self.upsample_rate, self.bit)
            mins.append(torch.tensor(lo, dtype=torch.float32, device=tensors[0].device))
            maxs.append(torch.tensor(hi, dtype=torch.float32, device=tensors[0].device))
        return mins, maxs

    def get_batch_tensors_qparams(self, act_tensors, alpha=0.01, args={}):
        """quant.py:561-586 -> (scales_list, zeros_list, qmin_list, qmax_list), one entry per module input."""
        if self.calib_algo == 'static_hist':
            assert self.sym is True and self.granularity == 'per_tensor', \
                'Only support per tensor static symmetric int quantize.'
            min_vals, max_vals = self.get_static_hist_range(act_tensors)
        elif self.calib_algo == 'static_minmax':
            min_vals, max_vals = self.get_static_minmax_range(act_tensors)
        elif self.calib_algo == 'static_moving_minmax':
            min_vals, max_vals = self.get_static_moving_minmax_range(act_tensors, alpha)
        else:
            raise ValueError(f'Unsupported calibration algorithm: {self.calib_algo}')
        scales_list, zeros_list, qmin_list, qmax_list = [], [], [], []
        for mn, mx in zip(min_vals, max_vals):
            scales, zeros, qmax, qmin = self.get_qparams((mn, mx), mn.device)
            scales_list.append(scales)
            zeros_list.append(zeros)
            qmin_list.append(qmin)
            qmax_list.append(qmax)
        return scales_list, zeros_list, qmin_list, qmax_list

    def _per_tensor_asym_qparams(self, tensor):
        """per_tensor + asymmetric (quant.py:132-136,555-556): min / max are 0-dim tensors of the tensor dtype and
        (qmax - qmin) is a 0-dim fp32 tensor, so type promotion makes scales and zeros fp32 0-dim. Four scalar ops
        on the device after one min/max pass — nothing to accelerate; written with the reference's dtypes."""
        _ffi.require_gpu(tensor)
        mn, mx = torch.aminmax(tensor)
        qmin, qmax = self.qmin.to(tensor.device), self.qmax.to(tensor.device)
        scales = (mx - mn).clamp(min=1e-5) / (qmax - qmin)
        if self.round_zp:
            zeros = (qmin - torch.round(mn / scales)).clamp(qmin, qmax)
        else:
            zeros = qmin - (mn / scales)
        return scales, zeros

    def _minmax_qparams(self, tensor):
        """tensor: reshaped, contiguous, on GPU. Returns (scales, zeros) in tensor dtype (fp32 for calib_algo mse and
        for per_tensor asymmetric)."""
        if self.calib_algo == 'mse':
            return self._mse_qparams(tensor)
        if self.granularity == 'per_tensor' and not self.sym:
            return self._per_tensor_asym_qparams(tensor)
        _ffi.require_gpu(tensor)
        L = _ffi.lib()
        tensor = tensor.contiguous()
        G, g = self._geometry(tensor)
        dt = _ffi.dt(tensor)
        scales = torch.empty(G, dtype=tensor.dtype, device=tensor.device)
        zeros = None if self.sym else torch.empty(G, dtype=tensor.dtype, device=tensor.device)
        ws = _ffi.workspace(L.llmc_minmax_qparams_ws_bytes(G, g), tensor.device)
        _ffi.check(L.llmc_minmax_qparams(
            _ffi.ptr(tensor), dt, G, g, int(self.sym), int(self.round_zp), float(self.qmin),
            float(self.qmax), _ffi.ptr(scales), _ffi.ptr(zeros), _ffi.ptr(ws), _ffi.stream()),
            'llmc_minmax_qparams')
        shp = self._qparam_shape(tensor)
        scales = scales.reshape(shp)
        # the reference returns torch.tensor(0.0) (0-dim fp32 on CPU) for symmetric zeros (quant.py:553)
        zeros = torch.tensor(0.0) if self.sym else zeros.reshape(shp)
        return scales, zeros


class IntegerQuantizer(BaseQuantizer):
    def __init__(self, bit, symmetric, granularity, **kwargs):
        super().__init__(bit, symmetric, granularity, **kwargs)
        self.quant_type = 'int-quant'
        if 'int_range' in self.kwargs:
            self.qmin = self.kwargs['int_range'][0]
            self.qmax = self.kwargs['int_range'][1]
        else:
            if self.sym:
                self.qmin = -(2 ** (self.bit - 1))
                self.qmax = 2 ** (self.bit - 1) - 1
            else:
                self.qmin = 0.0
                self.qmax = 2 ** self.bit - 1
        self.qmin = torch.tensor(self.qmin)
        self.qmax = torch.tensor(self.qmax)
        self.dst_nbins = 2 ** bit

    # ---- qparams ---------------------------------------------------------------------------------
    def get_tensor_qparams(self, tensor, args={}):
        """quant.py:690-697 -> (reshaped tensor, scales, zeros, qmax, qmin)."""
        tensor = self.reshape_tensor(tensor)
        if self.calib_algo == 'learnable' and (args.get('upbound_factor') is not None):
            scales, zeros, qmax, qmin = self.get_qparams(self.get_tensor_range(tensor, args), tensor.device)
            return tensor, scales, zeros, qmax, qmin
        scales, zeros = self._minmax_qparams(tensor)
        return tensor, scales, zeros, self.qmax.to(tensor.device), self.qmin.to(tensor.device)

    # ---- static arithmetic with given qparams (quant.py:699-717) ------------------------------------
    def _static(self, tensor, scales, zeros, qmax, qmin, out_kind):
        _ffi.require_gpu(tensor, scales)
        fz = 0 if self.round_zp else _ffi.FRACTIONAL_ZP     # quant.py:702-707: round(x / s.clamp_min(1e-9) + z)
        L = _ffi.lib()
        tensor = tensor.contiguous()
        g = tensor.shape[-1] if self.granularity != 'per_tensor' else tensor.numel()
        G = tensor.numel() // g
        # a 0-dim operand keeps its precision (opmath) but does not take part in type promotion against a dimensioned
        # tensor: every op still rounds to the tensor dtype (per_tensor qparams: fp32 0-dim scale, 16-bit weight)
        s_flag = _ffi.SCALAR_QPARAM if (scales.dim() == 0 and tensor.dim() > 0) else 0
        z_flag = _ffi.SCALAR_QPARAM if (torch.is_tensor(zeros) and zeros.dim() == 0 and tensor.dim() > 0) else 0
        scales_f = scales.reshape(-1)
        if scales_f.numel() == 1 and G > 1:
            scales_f = scales_f.expand(G)
        scales_f = scales_f.contiguous()
        if scales_f.numel() != G:
            raise ValueError(f'scales has {scales_f.numel()} entries, tensor has {G} groups')
        zt = None
        if torch.is_tensor(zeros) and zeros.dim() > 0 and zeros.numel() > 0:
            zt = zeros.reshape(-1)
            if zt.numel() == 1 and G > 1:
                zt = zt.expand(G)
            zt = zt.to(tensor.device).contiguous()
            if not zt.is_floating_point():
                zt = zt.to(scales_f.dtype)
        elif torch.is_tensor(zeros) and zeros.dim() == 0 and float(zeros) != 0.0:
            zdt = zeros.dtype if zeros.is_floating_point() else scales_f.dtype
            zt = torch.full((G,), float(zeros), dtype=zdt, device=tensor.device)
        elif not torch.is_tensor(zeros) and zeros not in (None, 0, 0.0):
            zt = torch.full((G,), float(zeros), dtype=scales_f.dtype, device=tensor.device)
        if out_kind == _ffi.OUT_FAKE:
            out = torch.empty_like(tensor)
        else:
            odt = {_ffi.OUT_I32: torch.int32, _ffi.OUT_I8: torch.int8, _ffi.OUT_U8: torch.uint8}[out_kind]
            out = torch.empty(tensor.shape, dtype=odt, device=tensor.device)
        _ffi.check(L.llmc_quant_static(
            _ffi.ptr(tensor), _ffi.dt(tensor), G, g, _ffi.ptr(scales_f), _ffi.dt(scales_f) | s_flag,
            _ffi.ptr(zt), ((_ffi.dt(zt) | z_flag) if zt is not None else 0) | fz, float(qmin), float(qmax), out_kind,
            _ffi.ptr(out), _ffi.stream()), 'llmc_quant_static')
        return out

    def _code_kind(self):
        if self.bit == 8:
            return _ffi.OUT_I8 if self.qmin != 0 else _ffi.OUT_U8
        return _ffi.OUT_I32

    def quant(self, tensor, scales, zeros, qmax, qmin):
        """Integer-valued codes in the promoted float dtype, like the reference's quant()."""
        codes = self._static(tensor, scales, zeros, qmax, qmin, _ffi.OUT_I32)
        return codes.to(torch.promote_types(tensor.dtype, scales.dtype))

    def dequant(self, tensor, scales, zeros):
        return (tensor - zeros.to(tensor.device) if torch.is_tensor(zeros) else tensor - zeros) * scales

    def quant_dequant(self, tensor, scales, zeros, qmax, qmin, output_scale_factor=1):
        if output_scale_factor != 1:
            raise NotImplementedError('output_scale_factor != 1 is outside the hot path')
        return self._static(tensor, scales, zeros, qmax, qmin, _ffi.OUT_FAKE)

    # ---- dynamic (fused min/max + quant) -----------------------------------------------------------
    def _dynamic(self, tensor, out_kind, want_qparams):
        _ffi.require_gpu(tensor)
        if self.granularity == 'per_tensor' and not self.sym and self.calib_algo != 'mse':
            tensor = tensor.contiguous()
            scales, zeros = self._per_tensor_asym_qparams(tensor)
            out = self._static(tensor, scales, zeros, self.qmax, self.qmin, out_kind)
            if not want_qparams:
                return out, None, None
            return out, scales, zeros
        if self.calib_algo == 'mse':   # searched range first, then the static arithmetic with its fp32 qparams
            tensor = tensor.contiguous()
            scales, zeros = self._mse_qparams(tensor)
            out = self._static(tensor, scales, zeros, self.qmax, self.qmin, out_kind)
            if not want_qparams:
                return out, None, None
            return out, scales.reshape(-1), None if self.sym else zeros.reshape(-1)
        L = _ffi.lib()
        tensor = tensor.contiguous()
        G, g = self._geometry(tensor)
        dt = _ffi.dt(tensor)
        if out_kind == _ffi.OUT_FAKE:
            out = torch.empty_like(tensor)
        else:
            odt = {_ffi.OUT_I32: torch.int32, _ffi.OUT_I8: torch.int8, _ffi.OUT_U8: torch.uint8}[out_kind]
            out = torch.empty(tensor.shape, dtype=odt, device=tensor.device)
        scales = zeros = None
        if want_qparams:
            scales = torch.empty(G, dtype=tensor.dtype, device=tensor.device)
            if not self.sym:
                zeros = torch.empty(G, dtype=tensor.dtype, device=tensor.device)
        ws = _ffi.workspace(L.llmc_quant_dynamic_ws_bytes(G, g), tensor.device)
        _ffi.check(L.llmc_quant_dynamic(
            _ffi.ptr(tensor), dt, G, g, int(self.sym), int(self.round_zp), float(self.qmin),
            float(self.qmax), out_kind, _ffi.ptr(out), _ffi.ptr(scales), _ffi.ptr(zeros), _ffi.ptr(ws),
            _ffi.stream()), 'llmc_quant_dynamic')
        return out, scales, zeros

    def fake_quant_weight_dynamic(self, weight, args={}):
        """quant.py:833-869"""
        if 'int_indices' in args or 'current_bit' in args:
            raise NotImplementedError('mixed int/fp columns and current_bit are outside the hot path')
        transpose = 'dim' in args and 'ic' in args['dim']
        q_weight = weight.T if transpose else weight
        org_w_shape = q_weight.shape
        if self.calib_algo == 'learnable' and args.get('upbound_factor') is not None:
            # quant.py:833-869 with the learnable range: qparams from the factor-scaled range, static arithmetic
            q_weight, scales, zeros, qmax, qmin = self.get_tensor_qparams(q_weight, args)
            q_weight = self.quant_dequant(q_weight.contiguous(), scales, zeros, qmax, qmin)
        else:
            q_weight = self.reshape_tensor(q_weight)
            q_weight, _, _ = self._dynamic(q_weight, _ffi.OUT_FAKE, False)
        q_weight = self.restore_tensor(q_weight, org_w_shape)
        return q_weight.T if transpose else q_weight

    def fake_quant_weight_static(self, weight, args):
        """quant.py:785-831"""
        if 'int_indices' in args or 'rounding' in args or 'output_scale_factor' in args:
            raise NotImplementedError('int_indices/rounding/output_scale_factor are outside the hot path')
        transpose = 'dim' in args and 'ic' in args['dim']
        q_weight = weight.T if transpose else weight
        org_w_shape = q_weight.shape
        org_w_dtype = q_weight.dtype
        q_weight = self.reshape_tensor(q_weight)
        q_weight = self.quant_dequant(q_weight, args['scales'], args['zeros'], args['qmax'], args['qmin'])
        q_weight = self.restore_tensor(q_weight, org_w_shape).to(org_w_dtype)
        return q_weight.T if transpose else q_weight

    def fake_quant_act_dynamic(self, act, args={}):
        """quant.py:753-783"""
        if 'int_indices' in args or 'current_bit' in args:
            raise NotImplementedError('mixed int/fp columns and current_bit are outside the hot path')
        org_shape = act.shape
        q_act = self.reshape_tensor(act)
        q_act, _, _ = self._dynamic(q_act, _ffi.OUT_FAKE, False)
        return self.restore_tensor(q_act, org_shape)

    def fake_quant_act_static(self, act, args={}):
        """quant.py:719-751"""
        org_shape, org_dtype = act.shape, act.dtype
        q_act = self.reshape_tensor(act)
        q_act = self.quant_dequant(q_act, args['scales'], args['zeros'], args['qmax'], args['qmin'])
        return self.restore_tensor(q_act, org_shape).to(org_dtype)

    def _finish_real(self, weight, scales, zeros):
        """common tail of real_quant_weight_* (quant.py:890-912)"""
        if self.granularity == 'per_tensor':
            qparams_shape = 1
        else:
            qparams_shape = (weight.shape[0], -1)
        if not self.sym and self.round_zp:
            zeros = zeros.to(weight.dtype)
        elif self.sym:
            zeros = None
        if zeros is not None:
            zeros = zeros.view(qparams_shape)
        scales = scales.view(qparams_shape)
        return weight, scales, zeros

    def real_quant_weight_static(self, weight, args):
        """quant.py:871-914 -> (int codes [R,K], scales [R,K/g], zeros [R,K/g] | None)"""
        if 'output_scale_factor' in args:
            raise NotImplementedError('output_scale_factor is outside the hot path')
        org_w_shape = weight.shape
        scales, zeros = args['scales'], args['zeros']
        w = self.reshape_tensor(weight)
        codes = self._static(w, scales, zeros, args['qmax'], args['qmin'], self._code_kind())
        codes = self.restore_tensor(codes, org_w_shape)
        return self._finish_real(codes, scales, zeros)

    def real_quant_weight_dynamic(self, weight, args={}):
        """quant.py:916-953"""
        org_w_shape = weight.shape
        w = self.reshape_tensor(weight)
        codes, scales, zeros = self._dynamic(w, self._code_kind(), True)
        codes = self.restore_tensor(codes, org_w_shape)
        if self.sym:
            zeros = torch.tensor(0.0)
        return self._finish_real(codes, scales, zeros)

    def __repr__(self):
        return (f'IntegerQuantizer(bit={self.bit}, sym={self.sym},'
                f'granularity={self.granularity},'
                f'kwargs={self.kwargs}, qmin={self.qmin}, qmax={self.qmax})')


def pack_lsb(codes, bits):
    """VllmRealQuantLinear.pack's integer part on the GPU (module_utils.py:836-862)."""
    _ffi.require_gpu(codes)
    L = _ffi.lib()
    codes = codes.contiguous()
    if codes.dtype == torch.int32:
        kind = _ffi.OUT_I32
    elif codes.dtype == torch.int8:
        kind = _ffi.OUT_I8
    else:
        raise ValueError(f'pack_lsb: codes must be int32 or int8, got {codes.dtype}')
    R, K = codes.shape
    pf = 32 // bits
    packed = torch.empty((R, (K + pf - 1) // pf), dtype=torch.int32, device=codes.device)
    _ffi.check(L.llmc_pack_lsb(_ffi.ptr(codes), kind, R, K, int(bits), _ffi.ptr(packed), _ffi.stream()),
               'llmc_pack_lsb')
    return packed


class FloatQuantizer(BaseQuantizer):
    """FP8 symmetric quantizer, e4m3 and e5m2: the `use_qtorch: True` path of the reference's FP8 configs
    (quant.py:963-1229). The rounding is `qtorch.quant.float_quantize(x, e_bits, m_bits, rounding='nearest')`
    (quant.py:1066-1068) — a third-party library the reference does not vendor (requirements/runtime.txt:29, no version pin)
    and this image does not have; its published algorithm (QPyTorch 0.3.0, quant_cpu.cpp / bit_helper.cpp) is restated in
    csrc/fp8_math.h and oracle/quant_ref.py: IEEE-style (e, m) formats with ties away from zero and saturation, so for
    e4m3 the largest code is 240 although qmax (finfo(float8_e4m3fn).max = 448) scales the tensor to +-448: every
    |x / scale| >= 248 lands on 240. `fp8_semantics='cast'` selects torch's own dtype cast instead (round to nearest even,
    OCP e4m3fn up to 448): what the reference's Triton kernels and its final `.to(torch.float8_e4m3fn)` compute."""

    _FMT = {'e4m3': (0, 4, 3, torch.float8_e4m3fn), 'e5m2': (1, 5, 2, torch.float8_e5m2)}

    def __init__(self, bit, symmetric, granularity, **kwargs):
        super().__init__(bit, symmetric, granularity, **kwargs)
        self.sym = True
        self.quant_type = 'float-quant'
        if self.bit not in self._FMT:
            raise NotImplementedError(f'FloatQuantizer bit={self.bit}: e4m3 and e5m2 are on the accelerated path '
                                      '(e3m2 / e4m7 / e2m1 of quant.py:988-990 are not 8-bit storage formats)')
        if self.granularity not in ('per_tensor', 'per_channel', 'per_token', 'per_group', 'per_block'):
            raise NotImplementedError(f'FloatQuantizer granularity={self.granularity}')
        self._fmt, self.e_bits, self.m_bits, self._tdtype = self._FMT[self.bit]
        if self.granularity == 'per_block' and self.bit != 'e4m3':
            raise NotImplementedError('FloatQuantizer per_block: e4m3 only (the DeepSeek-V3 checkpoint format)')
        self.use_qtorch = self.kwargs.get('use_qtorch')       # quant.py:974: absent means False, like the reference
        if not self.use_qtorch:
            raise NotImplementedError('FloatQuantizer without use_qtorch: True (get_float_qparams, quant.py:1005-1041: hard-coded '
                                      '.cuda(), per-element exponent scales) is outside the hot path')
        sem = self.kwargs.get('fp8_semantics', 'qtorch')
        if sem not in ('qtorch', 'cast'):
            raise ValueError(f"fp8_semantics must be 'qtorch' or 'cast', got {sem!r}")
        self._mode = (self._fmt << 4) | (0x100 if sem == 'qtorch' else 0)
        fmax = float(torch.finfo(self._tdtype).max)             # quant.py:985-1003
        self.qmax = torch.tensor(fmax)
        self.qmin = torch.tensor(-fmax)

    def _run(self, tensor, fake, scales=None):
        _ffi.require_gpu(tensor, scales)
        L = _ffi.lib()
        tensor = tensor.contiguous()
        G, g = self._geometry(tensor)
        sdtype = torch.float32 if self.granularity == 'per_tensor' else tensor.dtype
        static = scales is not None
        if static:
            s = scales.reshape(-1).to(tensor.device).contiguous()
            if s.numel() != G:
                raise ValueError(f'FloatQuantizer: {s.numel()} scales for {G} rows')
            sdtype = s.dtype
        else:
            s = torch.empty(G, dtype=sdtype, device=tensor.device)
        out = torch.empty_like(tensor) if fake else torch.empty(tensor.shape, dtype=torch.uint8, device=tensor.device)
        ws = None if static else _ffi.workspace(L.llmc_fp8_quant_ws_bytes(G, g), tensor.device)
        _ffi.check(L.llmc_fp8_quant(_ffi.ptr(tensor), _ffi.dt(tensor), G, g, int(bool(fake)) | self._mode, _ffi.ptr(out), _ffi.ptr(s),
                                    _ffi.dt(sdtype), int(static), _ffi.ptr(ws), _ffi.stream()), 'llmc_fp8_quant')
        return out, s.reshape(self._qparam_shape(tensor))

    # ---- per_block (quant.py:132-143, 612-658): b x b tiles of a 2-D weight, fp32 scales [M/b, 1, N/b, 1] ----------
    def _run_block(self, weight, fake, scales=None):
        _ffi.require_gpu(weight, scales)
        if weight.dim() != 2:
            raise ValueError('per_block quantization takes a 2-D weight')
        L = _ffi.lib()
        w = weight.contiguous()
        M, N = w.shape
        b = int(self.block_size)
        mb, nb = -(-M // b), -(-N // b)
        static = scales is not None
        s = (scales.reshape(mb, nb).to(device=w.device, dtype=torch.float32).contiguous() if static
             else torch.empty((mb, nb), dtype=torch.float32, device=w.device))
        out = torch.empty_like(w) if fake else torch.empty((M, N), dtype=torch.uint8, device=w.device)
        _ffi.check(L.llmc_fp8_block_quant(_ffi.ptr(w), _ffi.dt(w), M, N, b, 1e-5, int(bool(fake)) | (2 if static else 0) | (self._mode & 0x100),
                                          _ffi.ptr(out), _ffi.ptr(s), _ffi.stream()), 'llmc_fp8_block_quant')
        return out, s

    def get_tensor_qparams(self, tensor, args={}):
        if self.granularity == 'per_block':
            _, s = self._run_block(tensor, True)
            b = int(self.block_size)
            M, N = tensor.shape
            mb, nb = s.shape
            t = tensor
            if M % b or N % b:                      # the reference pads with zeros before viewing [M/b, b, N/b, b]
                t = torch.zeros((mb * b, nb * b), dtype=tensor.dtype, device=tensor.device)
                t[:M, :N] = tensor
            return t.view(mb, b, nb, b), s.view(mb, 1, nb, 1), torch.tensor(0.0), self.qmax, self.qmin
        tensor = self.reshape_tensor(tensor)
        _, scales = self._run(tensor, True)
        return tensor, scales, torch.tensor(0.0), self.qmax, self.qmin

    # ---- static arithmetic with given scales (quant.py:1061-1081) ----------------------------------------------------
    def quant(self, tensor, scales, zeros, qmax, qmin):
        """quant.py:1061-1072: float_quantize(tensor / scales + zeros) — values ON the 8-bit grid, in fp32 (the reference
        discards its cast back to the input dtype). `zeros` is 0 for every FloatQuantizer configuration (sym only)."""
        if torch.is_tensor(zeros) and zeros.numel() and bool((zeros != 0).any()) or (not torch.is_tensor(zeros) and zeros):
            raise NotImplementedError('FloatQuantizer.quant: non-zero zero points do not occur (the quantizer is symmetric)')
        if self.granularity == 'per_block' and tensor.dim() == 4:
            mb, b, nb, _ = tensor.shape
            bits, _ = self._run_block(tensor.reshape(mb * b, nb * b), False, scales=scales)
            return bits.view(self._tdtype).float().view(mb, b, nb, b)
        bits, _ = self._run(tensor, False, scales=scales)
        return bits.view(self._tdtype).float().reshape(tensor.shape)

    def dequant(self, tensor, scales, zeros):
        return (tensor - zeros.to(tensor.device) if torch.is_tensor(zeros) else tensor - zeros) * scales

    def quant_dequant(self, tensor, scales, zeros, qmax, qmin):
        return self.dequant(self.quant(tensor, scales, zeros, qmax, qmin), scales, zeros)

    def fake_quant_weight_dynamic(self, weight, args={}):
        if self.granularity == 'per_block':
            return self._run_block(weight, True)[0]
        out, _ = self._run(self.reshape_tensor(weight), True)
        return out.reshape(weight.shape)

    def fake_quant_act_dynamic(self, act, args={}):
        out, _ = self._run(self.reshape_tensor(act), True)
        return out.reshape(act.shape)

    def fake_quant_act_static(self, act, args={}):
        out, _ = self._run(self.reshape_tensor(act), True, scales=args['scales'])
        return out.reshape(act.shape)

    def fake_quant_weight_static(self, weight, args):
        if self.granularity == 'per_block':
            return self._run_block(weight, True, scales=args['scales'])[0]
        out, _ = self._run(self.reshape_tensor(weight), True, scales=args['scales'])
        return out.reshape(weight.shape)

    def _finish(self, bits, scales, shape):
        weight = bits.view(self._tdtype).reshape(shape)
        qshape = 1 if self.granularity == 'per_tensor' else (shape[0], -1)
        return weight, scales.reshape(qshape), None

    def real_quant_weight_dynamic(self, weight, args={}):
        if self.granularity == 'per_block':         # scales.view(scales.shape[0], scales.shape[2]) (quant.py:1215-1216)
            bits, s = self._run_block(weight, False)
            return bits.view(torch.float8_e4m3fn), s, None
        bits, scales = self._run(self.reshape_tensor(weight), False)
        return self._finish(bits, scales, weight.shape)

    def real_quant_weight_static(self, weight, args):
        if self.granularity == 'per_block':
            bits, s = self._run_block(weight, False, scales=args['scales'])
            return bits.view(torch.float8_e4m3fn), s, None
        bits, scales = self._run(self.reshape_tensor(weight), False, scales=args['scales'])
        return self._finish(bits, scales, weight.shape)

    def __repr__(self):
        return (f'FloatQuantizer(bit={self.bit},e_bits={self.e_bits}, m_bits={self.m_bits},'
                f'granularity={self.granularity},kwargs={self.kwargs}, qmin={self.qmin}, qmax={self.qmax})')


def weight_cast_to_bf16(weight, scale, block_size):
    """quant.py:18-30 (the non-Triton spelling of kernel.py's function): block-scaled e4m3 weight -> bf16."""
    from .kernel import weight_cast_to_bf16 as _cast
    return _cast(weight.contiguous(), scale.contiguous(), block_size, dtype=torch.bfloat16)


def weight_cast_to_fp8(weight, block_size, fp8_semantics='qtorch'):
    """quant.py:33-43: FloatQuantizer(e4m3, per_block).real_quant_weight_dynamic -> (fp8 weight, fp32 block scales).
    (kernel.py's function of the same name is the Triton kernel's arithmetic: the e4m3fn cast, no scale clamp.)"""
    q = FloatQuantizer(bit='e4m3', symmetric=True, granularity='per_block', block_size=block_size, use_qtorch=True,
                       fp8_semantics=fp8_semantics)
    w, s, _ = q.real_quant_weight_dynamic(weight)
    return w, s


def pack_awq_gemm(weight, scales, zeros, group_size):
    """AutoawqRealQuantLinear.gemm_pack (module_utils.py:1004-1065) on the GPU.
    weight [R,K] float, scales [R,K/g], zeros [R,K/g] int32 -> (qweight [K,R/8] i32, scales [K/g,R] f16, qzeros)."""
    _ffi.require_gpu(weight, scales, zeros)
    L = _ffi.lib()
    weight, scales = weight.contiguous(), scales.contiguous()
    zeros = zeros.to(torch.int32).contiguous()
    R, K = weight.shape
    ng = K // group_size
    qweight = torch.empty((K, R // 8), dtype=torch.int32, device=weight.device)
    qzeros = torch.empty((ng, R // 8), dtype=torch.int32, device=weight.device)
    sout = torch.empty((ng, R), dtype=torch.float16, device=weight.device)
    _ffi.check(L.llmc_pack_awq_gemm(_ffi.ptr(weight), _ffi.dt(weight), _ffi.ptr(scales), _ffi.dt(scales),
                                    _ffi.ptr(zeros), R, K, int(group_size), _ffi.ptr(qweight), _ffi.ptr(qzeros),
                                    _ffi.ptr(sout), _ffi.stream()), 'llmc_pack_awq_gemm')
    return qweight, sout, qzeros
