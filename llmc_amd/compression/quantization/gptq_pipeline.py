"""GPTQ for a set of Linear layers that share one input, on one GPU.

The reference handles every Linear separately (one Hessian, one factorisation and one column loop per
layer, gptq.py:97-117), although layers fed by the same activation (q/k/v, gate/up) have the same H.
Rows of W are independent given Hinv, so the layers of a subset are stacked along the output dimension
and go through ONE Hessian, ONE factorisation and ONE column loop; results are split per layer.
"""
from dataclasses import dataclass

import torch

from llmc_amd import _ffi

from . import gptq_ops
from .hessian import HessianAccumulator


@dataclass
class GptqConfig:
    bit: int = 4
    symmetric: bool = False
    group_size: int = 128          # 0 = per_channel
    actorder: bool = True
    static_groups: bool = False
    percdamp: float = 0.01
    blocksize: int = 128

    @property
    def qrange(self):
        if self.symmetric:
            return float(-(2 ** (self.bit - 1))), float(2 ** (self.bit - 1) - 1)
        return 0.0, float(2 ** self.bit - 1)


@dataclass
class GptqResult:
    weight: torch.Tensor           # [R, K] fp32, error-compensated (the reference leaves it fp32)
    scales: torch.Tensor           # [R, ng] fp32 (dynamic groups: processing order)
    zeros: torch.Tensor            # [R, ng] fp32 | None
    perm: torch.Tensor             # [K] int64 | None
    loss: torch.Tensor             # 0-dim fp32 on device: sum(Losses) (the reference logs it, gptq.py:184)
    info: torch.Tensor = None      # int32[1] on device: 0, or the leading minor at which the factorisation failed


def factor_from_hessian(H, cfg, h_work=None):
    """perm (actorder), then prep + factorisation of H. Returns (perm | None, U). H's dead diagonal is fixed
    in place like the reference does."""
    perm = None
    if cfg.actorder:
        perm = torch.argsort(torch.diagonal(H), descending=True)   # gptq.py:63
    Hp, _ = gptq_ops.hessian_prep(H, None, perm, cfg.percdamp, want_h=True, h_out=h_work)
    U = gptq_ops.chol_inv_upper(Hp, check=False)
    return perm, U


def _inverse_permutation(perm):
    """torch.argsort(perm) of gptq.py:186 for a permutation: one scatter instead of a radix sort (same indices)."""
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device, dtype=perm.dtype)
    return inv


def _prep_and_factor(H, W, perm, percdamp, h_work):
    """process_hessian_and_weights (gptq.py:128-176): dead fix, gather, damping, then U with H^-1 = U^T U. Round 5: the
    permuted matrix is gathered index-reversed (the same gather, the permutation read backwards) so that the factorisation
    starts from it directly — one pass over K^2 floats less on the chain of every subset (0.5 ms at K = 14336), same bits.
    The factor then lives in the calling stream's factorisation workspace: valid until that stream's next factorisation.
    _ffi.option(k3_fused_prep=0) keeps the two separate entry points."""
    if _ffi.HOST_OPTIONS['k3_fused_prep']:
        Hrev, Wp = gptq_ops.hessian_prep(H, W, perm, percdamp, want_h=True, h_out=h_work, reverse_h=True)
        U, info = gptq_ops.chol_inv_upper_rev(Hrev, check=False, return_info=True)
        return U, Wp, info
    Hp, Wp = gptq_ops.hessian_prep(H, W, perm, percdamp, want_h=True, h_out=h_work)
    U, info = gptq_ops.chol_inv_upper(Hp, check=False, return_info=True)
    return U, Wp, info


def quantize_stacked(W_list, H, cfg, static_qparams=None, h_work=None, want_losses=True, rows=None):
    """W_list: weights [R_i, K] (model dtype or fp32) of layers sharing the input whose Hessian is H.
    static_qparams: list of (scales [R_i, ng], zeros [R_i, ng] | None) in ORIGINAL column order, required
    when cfg.static_groups or per_channel. Returns a list of GptqResult.
    rows = (r0, r1): quantize only that row range of the stacked matrix (row-sharded multi-GPU mode: rows are
    independent given Hinv); the result is ONE GptqResult for the slice."""
    _ffi.require_gpu(H, *W_list)
    K = H.shape[0]
    Wcat = torch.cat([w.reshape(w.shape[0], -1) for w in W_list], dim=0) if len(W_list) > 1 else W_list[0]
    if rows is not None:
        if static_qparams is not None:
            sc = torch.cat([s.reshape(w.shape[0], -1) for (s, _), w in zip(static_qparams, W_list)], 0)[rows[0]:rows[1]]
            zr = None
            if static_qparams[0][1] is not None:
                zr = torch.cat([z.reshape(w.shape[0], -1) for (_, z), w in zip(static_qparams, W_list)], 0)[rows[0]:rows[1]]
            static_qparams = [(sc, zr)]
        W_list = [Wcat[rows[0]:rows[1]]]
        Wcat = W_list[0]
    rows = [w.shape[0] for w in W_list]
    # dead flags must come from the un-fixed diagonal, so W is gathered in the same call that fixes H
    perm = None
    if cfg.actorder:
        perm = torch.argsort(torch.diagonal(H), descending=True)
    U, Wp, info = _prep_and_factor(H, Wcat, perm, cfg.percdamp, h_work)   # callers check `info` at their next sync
    qmin, qmax = cfg.qrange
    static_mode = cfg.static_groups or not cfg.group_size
    scales = zeros = col_group = None
    if static_mode:
        scales = torch.cat([s.reshape(r, -1).float() for (s, _), r in zip(static_qparams, rows)], dim=0)
        if not cfg.symmetric:
            zeros = torch.cat([z.reshape(r, -1).float() for (_, z), r in zip(static_qparams, rows)], dim=0)
        if cfg.group_size:
            idx = perm if perm is not None else torch.arange(K, device=H.device)
            col_group = (idx // cfg.group_size).to(torch.int32)    # gptq.py:225-227
    tmp, losses, s, z = gptq_ops.gptq_quantize(Wp, U, cfg.symmetric, qmin, qmax, cfg.group_size,
                                               cfg.static_groups, col_group, scales, zeros,
                                               want_losses=want_losses, blocksize=cfg.blocksize)
    if perm is not None:
        invperm = _inverse_permutation(perm)
        K4 = tmp.shape[1]
        # gptq.py:188. LDS-staged gather where the shape allows it (K % 4 == 0, K <= 40960: a row of up to 160 KB staged in LDS)
        tmp = gptq_ops.gather_cols(tmp, invperm) if (K4 % 4 == 0 and K4 <= gptq_ops.GATHER_MAX_K) else tmp.index_select(1, invperm)
    out = []
    r0 = 0
    for r in rows:
        sl = slice(r0, r0 + r)
        out.append(GptqResult(weight=tmp[sl], scales=s[sl], zeros=None if z is None else z[sl], perm=perm,
                              loss=losses[sl].sum() if losses is not None else None, info=info))
        r0 += r
    return out


def owq_permutation(h_diag, n_out):
    """hessian_sorting with OWQ (gptq.py:66-83; OWQ forces actorder off): the non-outlier columns in their original
    order, then the n_out columns with the largest Hessian diagonal, largest first."""
    K = h_diag.shape[0]
    desc = torch.argsort(h_diag, descending=True)
    keep = torch.ones(K, dtype=torch.bool, device=h_diag.device)
    keep[desc[:n_out]] = False
    return torch.cat([torch.arange(K, device=h_diag.device)[keep], desc[:n_out]])


def quantize_owq(W, H, cfg, n_out, wquantizer, rtn_scales=None, rtn_zeros=None, h_work=None):
    """GPTQ + OWQ for ONE layer (gptq.py:44-56, 128-196): n_out outlier columns (largest Hessian diagonal) are moved
    last, kept in floating point and only receive the error feedback. Returns a GptqResult whose `weight` already has
    the compensated fp outlier columns in place and is back in the original column order; `n_nonout` rides in `extra`."""
    _ffi.require_gpu(H, W)
    K = H.shape[0]
    n_nonout = K - int(n_out)
    perm = owq_permutation(torch.diagonal(H), int(n_out))
    U, Wp, info = _prep_and_factor(H, W, perm, cfg.percdamp, h_work)
    qmin, qmax = cfg.qrange
    R = Wp.shape[0]
    if cfg.group_size:
        ng = K // cfg.group_size
        init_s = rtn_scales.reshape(R, ng) if rtn_scales is not None else None
        init_z = rtn_zeros.reshape(R, ng) if (rtn_zeros is not None and rtn_zeros.dim() > 0) else None
        tmp, losses, s, z = gptq_ops.gptq_quantize(Wp, U, cfg.symmetric, qmin, qmax, cfg.group_size, n_quant=n_nonout,
                                                   init_scales=init_s, init_zeros=init_z, blocksize=cfg.blocksize)
    else:
        # per_channel: qparams of the permuted, dead-zeroed non-outlier columns in fp32 (gptq.py:157-164)
        _, s, z, _, _ = wquantizer.get_tensor_qparams(Wp[:, :n_nonout].contiguous())
        tmp, losses, s, z = gptq_ops.gptq_quantize(Wp, U, cfg.symmetric, qmin, qmax, 0, scales=s,
                                                   zeros=None if cfg.symmetric else z, n_quant=n_nonout,
                                                   blocksize=cfg.blocksize)
    tmp[:, n_nonout:] = Wp[:, n_nonout:]                     # gptq.py:187: the compensated fp outlier columns
    invperm = _inverse_permutation(perm)
    K4 = tmp.shape[1]
    tmp = gptq_ops.gather_cols(tmp, invperm) if (K4 % 4 == 0 and K4 <= gptq_ops.GATHER_MAX_K) else tmp.index_select(1, invperm)
    return GptqResult(weight=tmp, scales=s, zeros=None if cfg.symmetric else z, perm=perm,
                      loss=losses.sum() if losses is not None else None, info=info)


def hessian_from_activations(X, acc=None):
    """X [n_seq, seq, K] (or [tokens, K]) 16-bit on the GPU -> H fp32 [K, K] with add_batch's scaling."""
    K = X.shape[-1]
    if acc is None:
        acc = HessianAccumulator(K, X.device)
    acc.reset()
    acc.add(X)
    return acc.H
