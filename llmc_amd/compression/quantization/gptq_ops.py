"""Functional wrappers over the GPTQ entry points of libllmc_hip.so (K2, K3, K4)."""
import torch

from llmc_amd import _ffi


def hessian_prep(H, W, perm, percdamp, want_h=True, h_out=None, reverse_h=False):
    """gptq.py:135-152,169-171. H [K,K] fp32 (dead diagonal fixed in place), W [R,K] any float dtype or None.
    Returns (Hout fp32 [K,K] permuted + damped | None, Wout fp32 [R,K] permuted, dead columns zeroed | None).
    reverse_h: Hout is written index-reversed (llmc_hessian_prep_rev) for chol_inv_upper_rev, which then needs no
    transposing pass in front of the factorisation."""
    _ffi.require_gpu(H, W, perm)
    L = _ffi.lib()
    K = H.shape[0]
    Hout = None
    if want_h:
        Hout = h_out if h_out is not None else torch.empty_like(H)
    Wout = None
    R = 0
    if W is not None:
        R = W.shape[0]
        W = W.contiguous()
        Wout = torch.empty((R, K), dtype=torch.float32, device=W.device)
    ws = _ffi.workspace(L.llmc_hessian_prep_ws_bytes(K), H.device)
    if perm is not None:
        perm = perm.to(torch.int64).contiguous()
    fn = L.llmc_hessian_prep_rev if reverse_h else L.llmc_hessian_prep
    _ffi.check(fn(_ffi.ptr(H), _ffi.ptr(W), _ffi.dt(W) if W is not None else 2, R, K,
                  _ffi.ptr(perm), float(percdamp), _ffi.ptr(Hout), _ffi.ptr(Wout), _ffi.ptr(ws),
                  _ffi.stream()), 'llmc_hessian_prep')
    return Hout, Wout


GATHER_MAX_K = 40960      # llmc_gather_cols stages one fp32 row in LDS (160 KB): K = 28672 (70B down_proj) included


def gather_cols(src, idx):
    """out[:, j] = src[:, idx[j]] for fp32 [R, K] on the GPU (the reference's `tmp[:, invperm]`, gptq.py:186-188)."""
    _ffi.require_gpu(src, idx)
    L = _ffi.lib()
    if src.dtype != torch.float32 or src.dim() != 2:
        raise ValueError('gather_cols: src must be fp32 [R, K]')
    src = src.contiguous()
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty_like(src)
    _ffi.check(L.llmc_gather_cols(_ffi.ptr(src), src.shape[0], src.shape[1], _ffi.ptr(idx), _ffi.ptr(out),
                                  _ffi.stream()), 'llmc_gather_cols')
    return out


_chol_ws = {}


def chol_inv_upper(H, check=True, return_info=False):
    """gptq.py:172-174 in one call: returns U upper with H^-1 = U^T U. H is overwritten.
    check=True synchronises and raises when H is not positive definite (torch.linalg.cholesky raises there);
    return_info=True hands back the device flag (0 = ok, k = leading minor k failed) for a deferred check."""
    _ffi.require_gpu(H)
    L = _ffi.lib()
    K = H.shape[0]
    need = L.llmc_chol_inv_upper_ws_bytes(K)
    key = (H.device, _ffi.stream())          # one workspace per stream: factorisations on different streams may overlap
    ws = _chol_ws.get(key)
    if ws is None or ws.numel() < need + 256:
        ws = torch.empty(need + 256, dtype=torch.uint8, device=H.device)
        _chol_ws[key] = ws
    off = (-ws.data_ptr()) % 256
    info = torch.zeros(1, dtype=torch.int32, device=H.device)
    _ffi.check(L.llmc_chol_inv_upper(_ffi.ptr(H), K, ws.data_ptr() + off, _ffi.ptr(info), _ffi.stream()),
               'llmc_chol_inv_upper')
    if check:
        i = int(info.item())
        if i != 0:
            raise RuntimeError(f'chol_inv_upper: matrix is not positive definite (leading minor {i})')
    return (H, info) if return_info else H


def chol_inv_upper_rev(Hrev, check=True, return_info=False, in_workspace=True):
    """chol_inv_upper for the index-reversed matrix of hessian_prep(..., reverse_h=True): Hrev is factored in place (destroyed) —
    bit-identical to chol_inv_upper on the un-reversed matrix, one K^2 pass less.
    ALIASING: by default (in_workspace=True) the returned U is a VIEW into the calling stream's factorisation workspace: it is
    overwritten by the next chol_inv_upper / chol_inv_upper_rev on the same stream and must not be read from another stream
    without an event. quantize_stacked / quantize_owq consume it at once. Callers that keep U (or factor a second Hessian before
    using the first) pass in_workspace=False and get a private tensor (one K^2 device copy)."""
    _ffi.require_gpu(Hrev)
    L = _ffi.lib()
    K = Hrev.shape[0]
    need = L.llmc_chol_inv_upper_ws_bytes(K)
    key = (Hrev.device, _ffi.stream())
    ws = _chol_ws.get(key)
    if ws is None or ws.numel() < need + 256:
        ws = torch.empty(need + 256, dtype=torch.uint8, device=Hrev.device)
        _chol_ws[key] = ws
    off = (-ws.data_ptr()) % 256
    info = torch.zeros(1, dtype=torch.int32, device=Hrev.device)
    # U lives in the (otherwise unused) work-matrix region of the per-stream workspace: valid until the next factorisation
    # on this stream, which is stream-ordered behind every kernel that reads it
    U = ws[off:off + K * K * 4].view(torch.float32).view(K, K)
    _ffi.check(L.llmc_chol_inv_upper_rev(_ffi.ptr(Hrev), _ffi.ptr(U), K, ws.data_ptr() + off, _ffi.ptr(info), _ffi.stream()),
               'llmc_chol_inv_upper_rev')
    if check:
        i = int(info.item())
        if i != 0:
            raise RuntimeError(f'chol_inv_upper: matrix is not positive definite (leading minor {i})')
    if not in_workspace:
        U = U.clone()
    return (U, info) if return_info else U


def raise_if_not_pd(info, what='Hessian'):
    """Deferred form of the check: one host sync. The reference's torch.linalg.cholesky raises on a non-PD matrix
    (bad calibration data, percdamp too small); silently continuing would write garbage weights."""
    i = int(info.item())
    if i != 0:
        raise RuntimeError(f'{what} is not positive definite (leading minor {i}): more calibration data or a larger '
                           'percdamp is needed (torch.linalg.cholesky raises here in the reference, gptq.py:172)')


def release_workspaces():
    _chol_ws.clear()


def gptq_quantize(W, Hinv, sym, qmin, qmax, group_size, static_groups=False, col_group=None, scales=None,
                  zeros=None, want_losses=True, blocksize=128, n_quant=None, init_scales=None, init_zeros=None):
    """gptq.py:199-244. W [R,K] fp32 (overwritten with the running weights), Hinv [K,K] fp32 upper.
    Returns (tmp [R,K], losses [R,K] | None, scales [R,ng], zeros [R,ng] | None).
    n_quant < K (OWQ): only the first n_quant columns are quantized, the rest keep receiving the error feedback and
    are left in W; tmp / losses are zero there and dynamic-group qparams of never-visited groups keep
    init_scales / init_zeros (the reference's `self.groups` starts from the layer's RTN qparams, gptq.py:380-395)."""
    _ffi.require_gpu(W, Hinv)
    L = _ffi.lib()
    R, K = W.shape
    per_channel = not group_size
    ng = 1 if per_channel else (K + group_size - 1) // group_size
    static_mode = static_groups or per_channel
    dev = W.device
    if static_mode:
        scales = scales.to(device=dev, dtype=torch.float32).reshape(R, ng).contiguous()
        if zeros is not None:
            zeros = zeros.to(device=dev, dtype=torch.float32).reshape(R, ng).contiguous()
        elif not sym:
            raise ValueError('gptq_quantize: zeros required for asymmetric static qparams')
    else:
        if init_scales is not None:
            scales = init_scales.to(device=dev, dtype=torch.float32).reshape(R, ng).contiguous().clone()
            zeros = (init_zeros.to(device=dev, dtype=torch.float32).reshape(R, ng).contiguous().clone()
                     if init_zeros is not None else torch.zeros((R, ng), dtype=torch.float32, device=dev))
        else:
            scales = torch.empty((R, ng), dtype=torch.float32, device=dev)
            zeros = torch.empty((R, ng), dtype=torch.float32, device=dev)
    if col_group is not None:
        col_group = col_group.to(device=dev, dtype=torch.int32).contiguous()
    nq = K if n_quant is None else int(n_quant)
    partial = nq < K
    tmp = torch.zeros_like(W) if partial else torch.empty_like(W)
    losses = (torch.zeros_like(W) if partial else torch.empty_like(W)) if want_losses else None
    ws = _ffi.workspace(L.llmc_gptq_quantize_ws_bytes(R, K), dev)
    _ffi.check(L.llmc_gptq_quantize_cols(
        _ffi.ptr(W), _ffi.ptr(Hinv), R, K, nq, int(bool(sym)), float(qmin), float(qmax), int(group_size or 0),
        int(bool(static_groups)), _ffi.ptr(col_group), _ffi.ptr(scales), _ffi.ptr(zeros), _ffi.ptr(tmp),
        _ffi.ptr(losses), int(blocksize), _ffi.ptr(ws), _ffi.stream()), 'llmc_gptq_quantize_cols')
    return tmp, losses, scales, zeros
