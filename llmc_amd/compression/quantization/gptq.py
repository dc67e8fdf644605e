"""GPTQ with llmc's operator surface (llmc/compression/quantization/gptq.py:21-478), arithmetic in HIP.

What is the same: constructor, `quant.special` keys (actorder, static_groups, percdamp, blocksize,
true_sequential, chunk_num), hook protocol (cache_input_hook -> add_batch), per-layer state in `layers_cache`,
module buffers buf_scales / buf_zeros / buf_perm / buf_invperm with the reference's shapes and dtypes (SURVEY.md
G2), fp32 `layer.weight.data` after the transform (G3), w_q / w_qdq / deploy / save_model.

What is different (mechanics, not arithmetic): layers of a subset that share their input share ONE Hessian,
ONE factorisation and ONE stacked column loop (the reference recomputes identical H and Hinv per layer); no
per-batch all_reduce of H — with several ranks (data-parallel calibration) H is reduced once per subset
(`_sync_hessian`), mathematically identical because every rank's running mean covers the same number of
sequences; no `.item()` sync per layer. OWQ (gptq.py:44-50, 66-83): the floating-point columns ride through the same
column loop (`quantize_owq`, llmc_gptq_quantize_cols).
"""
import copy
import math

import torch
import torch.distributed as dist

from llmc_amd.utils.registry_factory import ALGO_REGISTRY

from .base_blockwise_quantization import BaseBlockwiseQuantization, _world
from . import gptq_ops
from .gptq_pipeline import GptqConfig, owq_permutation, quantize_owq, quantize_stacked
from .hessian import HessianAccumulator
from .module_utils import _LLMC_LINEAR_TYPES_, _TRANSFORMERS_LINEAR_TYPES_


@ALGO_REGISTRY
class GPTQ(BaseBlockwiseQuantization):
    def __init__(self, model, quant_config, input, padding_mask, config, modality='language'):
        super().__init__(model, quant_config, input, padding_mask, config)
        self.model_dtype = next(self.model.model.parameters()).dtype
        self.add_quant_config()
        self.layers_cache = {}
        self._groups, self._group_of = {}, {}
        self.collect_model_qparams()

    @torch.no_grad()
    def add_quant_config(self):
        special = self.quant_config['special']
        self.true_sequential = special['true_sequential']
        self.static_groups = special['static_groups']
        self.actorder = special['actorder']
        self.percdamp = special['percdamp']
        self.blocksize = special['blocksize']
        self.chunk_num = special.get('chunk_num', 1)   # a memory lever of the reference's matmul; not needed here
        # not a reference key: False makes the Hessian accumulators copy every hooked sample instead of keeping a
        # reference to it until the subset's single launch (hessian.py)
        self.hessian_defer = bool(special.get('hessian_defer', True))
        # not a reference key: diag(H) — actorder's sort key, the damping mean — folded into fp64 inside the Hessian kernel (default;
        # hessian.py: exact_diag). False keeps the fp32 chain's own diagonal (rounds 1-5's default), for A/B
        self.hessian_exact_diag = bool(special.get('hessian_exact_diag', True))
        self.owq = bool(special.get('owq', False))
        if self.owq:                                   # gptq.py:47-50: OWQ fixes dynamic groups and no actorder
            self.n_outs = special['n_outs']
            self.static_groups = False
            self.actorder = False
        if self.wquantizer.calib_algo == 'mse' and not self.static_groups and self.wquantizer.granularity == 'per_group':
            # the column loop's kernel takes the qparams of a group from min/max of the current weights; searched
            # ranges inside the loop are not on the accelerated path (static_groups / per-channel use the quantizer)
            raise NotImplementedError('GPTQ with calib_algo=mse needs static_groups (or per_channel weights)')
        self.need_perm = (self.wquantizer.granularity == 'per_group' and not self.static_groups
                          and self.actorder) or self.owq
        gs = self.wquantizer.group_size if self.wquantizer.granularity == 'per_group' else 0
        self.gcfg = GptqConfig(bit=self.wquantizer.bit, symmetric=self.wquantizer.sym, group_size=gs,
                               actorder=self.actorder, static_groups=self.static_groups, percdamp=self.percdamp,
                               blocksize=self.blocksize)

    # ---- calibration: Hessian accumulation ---------------------------------------------------------
    @torch.no_grad()
    def cache_input_hook(self, m, inp, out, name, feat_dict):
        if isinstance(m, tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_)):
            self.add_batch(self.named_layers[name], name, inp[0].detach(), out.data)
        if self.act_static:
            super().cache_input_hook(m, inp, out, name, feat_dict)

    # ---- Hessian groups -----------------------------------------------------------------------------------------
    # The layers of a subset are set up as ONE group sharing one accumulator (q/k/v and gate/up see the same tensor;
    # the reference accumulates the identical H once per layer). The sharing is only KEPT for layers whose hooked
    # input is provably the very tensor the group's accumulator was fed with IN THE SAME FORWARD PASS of the block:
    #   * a pass is identified by `self._fwd_pass` (bumped by block_forward before every sample) and, for callers that
    #     drive add_batch themselves, by a member being called a second time;
    #   * the feeder's tensor is kept referenced for the duration of the pass, so its storage cannot be handed to
    #     another tensor that would then alias its signature;
    #   * a member that sees a different tensor, or whose first call comes after the group has already been fed
    #     (an expert that received no token on earlier samples: HF / DeepSeek forwards skip it), leaves the group and
    #     accumulates its own Hessian exactly like the reference — the experts and the router of a Mixtral / Qwen2-MoE /
    #     DeepSeek subset each see their own routed tokens (mixtral.py:65-66, qwen2moe.py:81-84, deepseekv2.py:130-135);
    #   * a member that DID share on earlier passes and then diverges cannot be repaired: loud failure.
    @staticmethod
    def _input_signature(inp):
        """What makes two hooked inputs THE SAME tensor (not merely equal): storage, view geometry, version."""
        return (inp.data_ptr(), tuple(inp.shape), tuple(inp.stride()), inp.dtype, inp._version)

    def _new_group(self, names, K, device):
        gid = self._next_gid = getattr(self, '_next_gid', 0) + 1
        acc = HessianAccumulator(K, device, defer=getattr(self, 'hessian_defer', True),
                                 exact_diag=getattr(self, 'hessian_exact_diag', True))
        self._groups[gid] = {'acc': acc, 'pass': None, 'passes': 0, 'members': list(names)}
        for n in names:
            self._group_of[n] = gid
            # 'H' is the accumulator's buffer; read `acc.H` to have the pending samples folded in first
            self.layers_cache[n] = {'acc': acc, 'H': getattr(acc, '_H', None), 'nsamples': 0, 'columns': K,
                                    'calls': 0, 'joined': 0, 'device': device}
        return gid

    def _leave_group(self, name, device):
        """`name` stops sharing: a group (accumulator) of its own, its call count kept."""
        c = self.layers_cache[name]
        g = self._groups[self._group_of[name]]
        if c['joined']:
            raise RuntimeError(f'GPTQ: layer {name} shared its Hessian on earlier calibration calls but now sees a '
                               'different input tensor than its subset; per-layer Hessians cannot be recovered')
        g['members'].remove(name)
        calls = c['calls']
        self._new_group([name], c['columns'], device)
        self.layers_cache[name]['calls'] = calls
        return self._groups[self._group_of[name]]

    def _feed(self, g, name, inp):
        g['acc'].add(inp)
        g['passes'] += 1
        for m in g['members']:
            self.layers_cache[m]['nsamples'] = g['acc'].nsamples
        self.layers_cache[name]['joined'] += 1

    @torch.no_grad()
    def add_batch(self, layer, name, inp, out):
        """gptq.py:254-295 (running-mean Hessian of the layer's input), fed once per distinct input tensor."""
        active = getattr(self, '_active_layers', None)
        if active is not None and name not in active:
            # true_sequential: the first pass over the block hooks every layer, but the Hessians of the later subsets
            # are re-initialised and re-accumulated after the earlier subsets have been quantized
            # (base_blockwise_quantization.py:506-526, gptq.py:311-315) — accumulating them now is wasted work
            # (for a Llama block: 3 of the 4 distinct Hessians, the widest one included)
            return
        c = self.layers_cache[name]
        c['calls'] += 1
        g = self._groups[self._group_of[name]]
        if len(g['members']) == 1:
            return self._feed(g, name, inp)
        dev = layer.weight.device
        token = getattr(self, '_fwd_pass', None)
        sig = self._input_signature(inp)
        p = g['pass']
        new_pass = p is None or name in p['seen'] or (token is not None and p['token'] != token)
        if not new_pass:
            if sig == p['sig']:                                # the very same tensor is already in H
                p['seen'].add(name)
                c['joined'] += 1
                c['nsamples'] = g['acc'].nsamples
                return
            return self._feed(self._leave_group(name, dev), name, inp)   # another tensor in the same pass: its own Hessian
        if c['joined'] != g['passes']:                         # it missed passes the group has been fed with
            return self._feed(self._leave_group(name, dev), name, inp)
        g['pass'] = {'token': token, 'sig': sig, 'ref': inp, 'seen': {name}}
        self._feed(g, name, inp)

    def _settle_group(self, gid):
        """Before a group's Hessian is used: the pass's reference is dropped and members that never took part leave
        (the reference's Hessian of a layer that was never called is all zero); partial sharers cannot be repaired."""
        g = self._groups[gid]
        g['pass'] = None
        for n in list(g['members']):
            c = self.layers_cache[n]
            if len(g['members']) > 1 and c['joined'] != g['passes']:
                self._leave_group(n, c['device'])

    def _group_layers(self, named_layers, block=None):
        """lists of layer names that start out sharing a Hessian: the model's subsets (same-shaped inputs only)."""
        groups, seen = [], set()
        if block is not None:
            for subset in self.model.get_subsets_in_block(block):
                by_k = {}
                for n in subset['layers']:
                    if n in named_layers and n not in seen:
                        by_k.setdefault(self._in_features(named_layers[n]), []).append(n)
                        seen.add(n)
                groups.extend(by_k.values())
        groups.extend([n] for n in named_layers if n not in seen)
        return groups

    @staticmethod
    def _in_features(layer):
        return layer.weight.shape[1] if layer.weight.dim() == 2 else layer.weight[0].numel()

    @torch.no_grad()
    def layer_init(self, layer, name):
        self._new_group([name], self._in_features(layer), layer.weight.device)

    @torch.no_grad()
    def subset_init(self, subset):
        self.named_layers = subset['layers']
        self._active_layers = set(subset['layers']) if getattr(self, 'true_sequential', False) else None
        by_k = {}
        for n, l in self.named_layers.items():
            by_k.setdefault(self._in_features(l), []).append(n)
        for K, names in by_k.items():
            self._new_group(names, K, self.named_layers[names[0]].weight.device)

    @torch.no_grad()
    def block_init(self, block):
        self.named_layers = self.model.get_block_linears(block)
        self._active_layers = None
        if getattr(self, 'true_sequential', False):
            subsets = self.model.get_subsets_in_block(block)
            if subsets:
                self._active_layers = set(subsets[0]['layers'])
        for names in self._group_layers(self.named_layers, block):
            l0 = self.named_layers[names[0]]
            self._new_group(names, self._in_features(l0), l0.weight.device)

    def _sync_hessian(self, gid):
        """One reduction per Hessian at the end of accumulation (the reference all-reduces per batch, gptq.py:292)."""
        g = self._groups[gid]
        if _world() > 1 and not g.get('synced'):
            H = g['acc'].H
            dist.all_reduce(H, op=dist.ReduceOp.SUM)
            H.div_(_world())
            g['synced'] = True

    # ---- transform ---------------------------------------------------------------------------------
    @torch.no_grad()
    def subset_transform(self, subset, input_feat, subset_kwargs):
        layers_dict = {n: l for n, l in subset['layers'].items()
                       if isinstance(l, tuple(_LLMC_LINEAR_TYPES_ + _TRANSFORMERS_LINEAR_TYPES_))}
        if self._overrides_reference_hooks():        # a subclass written against the reference's per-layer methods
            for n, l in layers_dict.items():
                self.layer_transform_reference(l, n)
                self.free(n)
            return
        for gid in {self._group_of[n] for n in layers_dict}:
            self._settle_group(gid)
        by_group = {}
        for n in layers_dict:
            by_group.setdefault(self._group_of[n], []).append(n)
        for gid, names in by_group.items():
            self._sync_hessian(gid)
            self._transform_group(gid, [layers_dict[n] for n in names], names)
            for n in names:
                self.free(n)

    @torch.no_grad()
    def layer_transform(self, layer, name):
        if self._overrides_reference_hooks():
            return self.layer_transform_reference(layer, name)
        self._settle_group(self._group_of[name])
        gid = self._group_of[name]
        self._sync_hessian(gid)
        self._transform_group(gid, [layer], [name])

    owq_permutation = staticmethod(owq_permutation)

    @torch.no_grad()
    def block_transform(self, block, input_feat, block_kwargs):
        if self.owq and not hasattr(self, 'n_out_dict'):       # gptq.py:89-93: n_outs follow get_block_linears' order
            self.n_out_dict = {n: self.n_outs[i] for i, n in enumerate(self.model.get_block_linears(block))}
        if not getattr(self, 'true_sequential', False) and _world() == 1:
            # one block forward has fed every subset's accumulator: the Hessians of one width (q|k|v, o and gate|up of a Llama
            # block) are formed by ONE launch (HessianAccumulator.flush_many) instead of one launch per subset
            flush_many = getattr(HessianAccumulator, 'flush_many', None)
            if flush_many is not None:
                flush_many([g['acc'] for g in self._groups.values()])
        super().block_transform(block, input_feat, block_kwargs)

    def _transform_owq(self, gid, layers, names):
        """OWQ: every layer has its own outlier count, hence its own permutation and factorisation (the Hessian is
        still shared by layers that see the same input)."""
        H = self._groups[gid]['acc'].H
        self.last_losses = {}
        for l, n in zip(layers, names):
            n_out = int(self.n_out_dict[n])
            z = l.buf_zeros if (torch.is_tensor(l.buf_zeros) and l.buf_zeros.dim() > 0) else None
            r = quantize_owq(l.weight.data, H.clone() if len(layers) > 1 else H, self.gcfg, n_out, self.wquantizer,
                             rtn_scales=l.buf_scales, rtn_zeros=z)
            gptq_ops.raise_if_not_pd(r.info, f'GPTQ/OWQ: Hessian of {n}')
            l.weight.data = r.weight.reshape(l.weight.shape)
            self.last_losses[n] = r.loss
            l.register_buffer('buf_perm', r.perm)
            l.register_buffer('buf_invperm', torch.argsort(r.perm))
            l.register_buffer('buf_n_nonout', torch.tensor(l.weight.shape[1] - n_out))
            if self.wquantizer.granularity == 'per_group':
                l.buf_scales = r.scales.reshape(-1, 1).clone()
                if not self.wquantizer.sym:
                    l.buf_zeros = r.zeros.reshape(-1, 1).clone()
            else:
                l.buf_scales = r.scales
                if not self.wquantizer.sym:
                    l.buf_zeros = r.zeros

    def _transform_group(self, gid, layers, names):
        if self.owq:
            return self._transform_owq(gid, layers, names)
        H = self._groups[gid]['acc'].H
        static = None
        if self.gcfg.static_groups or not self.gcfg.group_size:
            static = []
            for l in layers:   # RTN qparams of the ORIGINAL weights, original column order
                z = l.buf_zeros if (torch.is_tensor(l.buf_zeros) and l.buf_zeros.dim() > 0) else None
                static.append((l.buf_scales, z))
        results = quantize_stacked([l.weight.data for l in layers], H, self.gcfg, static_qparams=static)
        # one host sync per subset (the reference has one per layer: `.item()` of the loss, gptq.py:184): a Hessian that
        # is not positive definite must raise like torch.linalg.cholesky does, not write garbage into the layer
        gptq_ops.raise_if_not_pd(results[0].info, 'GPTQ: Hessian of ' + ', '.join(names))
        self.last_losses = {}
        for l, n, r in zip(layers, names, results):
            l.weight.data = r.weight.reshape(l.weight.shape)          # fp32, like the reference (G3)
            self.last_losses[n] = r.loss
            if self.actorder:
                l.register_buffer('buf_perm', r.perm)
                l.register_buffer('buf_invperm', torch.argsort(r.perm))
            if self.wquantizer.granularity == 'per_group' and not self.static_groups:
                l.buf_scales = r.scales.reshape(-1, 1).clone()        # merge_qparams: [R * K/g, 1] fp32
                if not self.wquantizer.sym:
                    l.buf_zeros = r.zeros.reshape(-1, 1).clone()

    # ================================================================================================================
    # The reference's per-layer method surface (gptq.py:58-83, 119-244, 333-409) — same names, arguments, return values
    # and in-place effects, arithmetic on the same HIP entry points the stacked path uses. `main()` never needs them
    # (subset_transform above runs a whole subset through ONE Hessian / factor / column loop); they exist for callers
    # and subclasses written against llmc's class: a subclass that overrides any of them is routed through the
    # reference's per-layer flow (`layer_transform_reference`) so that its override is honoured.
    # ================================================================================================================
    _REFERENCE_HOOKS = ('hessian_sorting', 'initialize_qparams_and_prepare_weights', 'process_hessian_and_weights',
                        'update_layer_with_transformed_weights', 'weight_transform', 'search_column_qparams',
                        'search_layer_qparams', 'search_group_qparams', 'split_qparams', 'merge_qparams',
                        'update_model_qparams', 'ready')

    def _overrides_reference_hooks(self):
        """True when a class outside this package (a maintainer's subclass) defines one of the reference-named hooks."""
        for h in self._REFERENCE_HOOKS:
            f = getattr(type(self), h, None)
            if f is not None and not (getattr(f, '__module__', '') or '').startswith('llmc_amd.'):
                return True
        return False

    def _hessian_of(self, name):
        """The layer's Hessian with every pending sample folded in (and reduced over the ranks)."""
        gid = self._group_of[name]
        self._settle_group(gid)
        self._sync_hessian(gid)
        return self._groups[self._group_of[name]]['acc'].H

    def hessian_sorting(self, name):
        """gptq.py:58-83: `self.perm` — actorder: columns by descending diag(H); OWQ: the non-outlier columns in their
        original order, then the n_out columns with the largest diagonal."""
        H = self._hessian_of(name)
        if not self.owq:
            if self.actorder:
                self.perm = torch.argsort(torch.diag(H), descending=True)
            return
        self.perm = owq_permutation(torch.diagonal(H), int(self.n_out))

    def initialize_qparams_and_prepare_weights(self, layer, name):
        """gptq.py:119-126."""
        self.qparams = {}
        self.columns = self.layers_cache[name]['columns']
        self.n_out = int(self.n_out_dict[name]) if self.owq else 0
        self.n_nonout = self.columns - self.n_out
        self.dev = layer.weight.device
        if self.actorder or self.owq:
            self.hessian_sorting(name)

    def process_hessian_and_weights(self, layer, name):
        """gptq.py:128-176 -> (W, H): W fp32 [R, K] with the dead columns zeroed and the columns permuted, H the upper
        factor U with (H_perm + damp I)^-1 = U^T U (llmc_hessian_prep + llmc_chol_inv_upper: one reverse Cholesky and one
        triangular inverse instead of cholesky -> cholesky_inverse -> cholesky(upper)). Raises like torch.linalg.cholesky
        when the damped Hessian is not positive definite. The shared accumulator keeps its Hessian (the reference deletes
        the per-layer copy): other layers of the subset read the same matrix."""
        W = layer.weight.data
        if W.dim() > 2:
            W = W.flatten(1)
        elif type(layer).__name__ == 'Conv1D':
            W = W.t()
        H = self._hessian_of(name)
        if not self.ready():
            if self.wquantizer.granularity == 'per_group':
                self.groups = []
                self.search_group_qparams(layer)
            else:
                self.search_layer_qparams(layer)
        perm = self.perm if (self.actorder or self.owq) else None
        Hp, Wp = gptq_ops.hessian_prep(H, W.contiguous(), perm, self.percdamp, want_h=True)
        if perm is not None:
            self.invperm = torch.argsort(self.perm)
            layer.register_buffer('buf_perm', self.perm)
            layer.register_buffer('buf_invperm', self.invperm)
            if self.owq:
                layer.register_buffer('buf_n_nonout', torch.tensor(self.n_nonout))
                if self.wquantizer.granularity == 'per_channel':
                    _, layer.buf_scales, layer.buf_zeros, _, _ = self.wquantizer.get_tensor_qparams(
                        Wp[:, :self.n_nonout].contiguous())
                    self.qparams['scale'], self.qparams['zero'] = layer.buf_scales, layer.buf_zeros
        U = gptq_ops.chol_inv_upper(Hp, check=True)
        return Wp, U

    def update_layer_with_transformed_weights(self, layer, W, H, name):
        """gptq.py:178-197."""
        Losses = torch.zeros_like(W)
        tmp = torch.zeros_like(W)
        self.weight_transform(W, H, Losses, tmp)
        self.last_losses = getattr(self, 'last_losses', {})
        self.last_losses[name] = Losses.sum()
        if self.actorder or self.owq:
            tmp[:, self.n_nonout:] = W[:, self.n_nonout:]
            tmp = gptq_ops.gather_cols(tmp, self.invperm) if (tmp.shape[1] % 4 == 0 and tmp.shape[1] <= gptq_ops.GATHER_MAX_K) \
                else tmp[:, self.invperm]
        if type(layer).__name__ == 'Conv1D':
            tmp = tmp.t()
        layer.weight.data = tmp.reshape(layer.weight.shape)
        if self.wquantizer.granularity == 'per_group' and not self.static_groups:
            self.update_model_qparams(layer)

    @torch.no_grad()
    def weight_transform(self, W, Hinv, Losses, tmp):
        """gptq.py:199-244, the blocked column loop (llmc_gptq_quantize_cols). In place like the reference: `tmp` receives
        the error-compensated weight of every visited column, `Losses` its loss, `W` the running weights — its columns
        from n_nonout on (OWQ's floating-point columns) end with every block's feedback applied, which is what the caller
        reads (gptq.py:187); the visited columns of W are scratch afterwards (the reference leaves each block's columns at
        their value before the block was entered). Uses self.n_nonout / columns / perm / qparams / groups as the reference
        does: dynamic groups (per_group without static_groups) write `self.groups[g]` and `self.qparams` at every group
        start (search_column_qparams, gptq.py:359-366), static groups read `self.groups`, per_channel reads `self.qparams`."""
        wq = self.wquantizer
        per_group = wq.granularity == 'per_group'
        gs = int(wq.group_size) if per_group else 0
        R, K = W.shape
        n_nonout = int(getattr(self, 'n_nonout', K))
        qmin, qmax = float(wq.qmin), float(wq.qmax)
        dynamic = per_group and not self.static_groups
        col_group = scales = zeros = init_s = init_z = None
        if dynamic:
            groups = getattr(self, 'groups', None)
            if groups and n_nonout < K:      # OWQ: groups the loop never visits keep their RTN qparams (gptq.py:380-395)
                init_s = torch.cat([g['scale'].reshape(R, 1).float() for g in groups], 1)
                if not wq.sym:
                    init_z = torch.cat([g['zero'].reshape(R, 1).float() for g in groups], 1)
        elif per_group:
            scales = torch.cat([g['scale'].reshape(R, 1).float() for g in self.groups], 1)
            if not wq.sym:
                zeros = torch.cat([g['zero'].reshape(R, 1).float() for g in self.groups], 1)
            idx = self.perm if self.actorder else torch.arange(K, device=W.device)
            col_group = (idx // gs).to(torch.int32)
        else:
            scales = self.qparams['scale'].reshape(R, 1).float()
            if not wq.sym:
                zeros = self.qparams['zero'].reshape(R, 1).float()
        if not W.is_contiguous():
            raise ValueError('GPTQ.weight_transform: W must be contiguous (it is updated in place)')
        t, l, s, z = gptq_ops.gptq_quantize(W, Hinv.contiguous(), wq.sym, qmin, qmax, gs, self.static_groups, col_group, scales,
                                            zeros, want_losses=True, blocksize=self.blocksize,
                                            n_quant=n_nonout if n_nonout < K else None, init_scales=init_s, init_zeros=init_z)
        tmp.copy_(t)
        Losses.copy_(l)
        if dynamic:
            ng = s.shape[1]
            if not isinstance(getattr(self, 'groups', None), list) or len(self.groups) < ng:
                self.groups = list(getattr(self, 'groups', None) or []) + [None] * (ng - len(getattr(self, 'groups', None) or []))
            qmax_t, qmin_t = wq.qmax.to(W.device), wq.qmin.to(W.device)
            for g in range(-(-n_nonout // gs)):          # the groups the loop visited
                self.groups[g] = {'scale': s[:, g:g + 1].clone(), 'zero': torch.tensor(0.0) if wq.sym else z[:, g:g + 1].clone(),
                                  'qmax': qmax_t, 'qmin': qmin_t}
            if n_nonout > 0:
                self.qparams = dict(self.groups[-(-n_nonout // gs) - 1])

    @torch.no_grad()
    def layer_transform_reference(self, layer, name):
        """gptq.py:113-117: the reference's per-layer flow through the methods above."""
        self.initialize_qparams_and_prepare_weights(layer, name)
        W, H = self.process_hessian_and_weights(layer, name)
        self.update_layer_with_transformed_weights(layer, W, H, name)

    # ---- qparam bookkeeping (gptq.py:333-409) ---------------------------------------------------------------------------
    @torch.no_grad()
    def split_qparams(self, qparams):
        group_num = math.ceil(self.columns / self.wquantizer.group_size)
        qparams = qparams.reshape(math.ceil(qparams.shape[0] / group_num), -1).t()
        return [q.reshape(-1, 1) for q in torch.split(qparams, 1, dim=0)]

    @torch.no_grad()
    def merge_qparams(self, qparams):
        if isinstance(qparams, int):
            return qparams
        if self.wquantizer.granularity == 'per_head':
            head_size = self.rows // self.head_num
            qparams = qparams.t().repeat(head_size, 1).t().reshape(-1, 1)
        elif self.wquantizer.granularity == 'per_group':
            qparams = torch.stack(qparams, dim=1).reshape(-1, 1)
        return qparams

    @torch.no_grad()
    def search_column_qparams(self, c_tensor, idx):
        """gptq.py:359-366: min/max qparams of one column group (llmc_minmax_qparams), kept as `self.qparams` and in
        `self.groups[idx // group_size]`."""
        _, scale, zero, qmax, qmin = self.wquantizer.get_tensor_qparams(c_tensor.contiguous())
        self.qparams = {'scale': scale, 'zero': zero, 'qmax': qmax, 'qmin': qmin}
        self.groups[idx // self.wquantizer.group_size] = copy.deepcopy(self.qparams)

    @torch.no_grad()
    def search_layer_qparams(self, layer):
        scales, zeros = self.merge_qparams(layer.buf_scales), layer.buf_zeros
        if not self.wquantizer.sym:
            zeros = self.merge_qparams(zeros)
        self.qparams['scale'], self.qparams['zero'] = scales, zeros
        self.qparams['qmax'], self.qparams['qmin'] = layer.buf_qmax, layer.buf_qmin

    @torch.no_grad()
    def search_group_qparams(self, layer):
        self.group_scales = self.split_qparams(layer.buf_scales)
        if not self.wquantizer.sym:
            self.group_zeros = self.split_qparams(layer.buf_zeros)
        for i in range(len(self.group_scales)):
            self.groups.append({'scale': self.group_scales[i],
                                'zero': self.group_zeros[i] if not self.wquantizer.sym else torch.tensor(0.0),
                                'qmax': layer.buf_qmax, 'qmin': layer.buf_qmin})

    @torch.no_grad()
    def update_model_qparams(self, layer):
        layer.buf_scales = copy.deepcopy(self.merge_qparams([g['scale'] for g in self.groups]))
        if not self.wquantizer.sym:
            layer.buf_zeros = copy.deepcopy(self.merge_qparams([g['zero'] for g in self.groups]))

    @torch.no_grad()
    def ready(self):
        if 'scale' not in self.qparams:
            return False
        return torch.all(self.qparams['scale'] != 0)

    @torch.no_grad()
    def collect_model_qparams(self):
        for block in self.blocks:
            block = block.cuda()
            self.collect_block_qparams(block)
            block = block.cpu()

    # ---- deploy-time quantization with the stored qparams (gptq.py:412-452) ---------------------------
    @torch.no_grad()
    def w_q(self, module, wquantizer):
        args = {'scales': module.buf_scales.to(self.model_dtype), 'zeros': module.buf_zeros,
                'qmax': module.buf_qmax, 'qmin': module.buf_qmin}
        return wquantizer.real_quant_weight_static(module.weight.data, args)

    @torch.no_grad()
    def w_qdq(self, module, wquantizer):
        weight = module.weight
        if self.need_perm:
            weight = module.weight[:, module.buf_perm]
        args = {'scales': module.buf_scales, 'zeros': getattr(module, 'buf_zeros', None),
                'qmax': module.buf_qmax, 'qmin': module.buf_qmin}
        owq = getattr(self, 'owq', False)
        if owq:
            fp_weight = weight[:, int(module.buf_n_nonout):]
        weight = wquantizer.fake_quant_weight_static(weight, args).to(self.model_dtype)
        if owq:                                            # gptq.py:441-447: outlier columns stay floating point
            weight[:, int(module.buf_n_nonout):] = fp_weight.to(self.model_dtype)
        if self.need_perm:
            weight = weight[:, module.buf_invperm]
        return weight

    @torch.no_grad()
    def deploy(self, quant_format):
        if quant_format not in ['fake_quant', 'origin_float']:
            assert not self.need_perm
        super().deploy(quant_format)
        self.model.convert_dtype(self.model_dtype)

    @torch.no_grad()
    def save_model(self, path):
        self.model.convert_dtype(self.model_dtype)
        super().save_model(path)

    @torch.no_grad()
    def free(self, name):
        self.layers_cache.pop(name, None)
        gid = self._group_of.pop(name, None)
        if gid is not None and gid in self._groups:
            g = self._groups[gid]
            if name in g['members']:
                g['members'].remove(name)
            if not g['members']:
                del self._groups[gid]
