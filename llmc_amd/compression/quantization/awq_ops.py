"""Functional wrappers over the AWQ entry points of libllmc_hip.so (K8, K9)."""
import os
import sys

import torch

from llmc_amd import _ffi


def act_mean(x):
    """Awq.get_act_scale (awq.py:74-76): x.abs().view(-1, K).mean(0) in the tensor dtype."""
    _ffi.require_gpu(x)
    L = _ffi.lib()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    N, K = x2.shape
    out = torch.empty(K, dtype=x.dtype, device=x.device)
    ws = _ffi.workspace(L.llmc_awq_act_mean_ws_bytes(N, K), x.device)
    _ffi.check(L.llmc_awq_act_mean(_ffi.ptr(x2), _ffi.dt(x2), N, K, _ffi.ptr(out), _ffi.ptr(ws), _ffi.stream()),
               'llmc_awq_act_mean')
    return out


def weight_mean(w, group_size):
    """one layer's `layer_scale.mean(0)` of Awq.get_weight_scale (awq.py:59-66)."""
    _ffi.require_gpu(w)
    L = _ffi.lib()
    w = w.contiguous()
    R, K = w.shape
    out = torch.empty(K, dtype=w.dtype, device=w.device)
    ws = _ffi.workspace(L.llmc_awq_weight_mean_ws_bytes(R, K), w.device)
    _ffi.check(L.llmc_awq_weight_mean(_ffi.ptr(w), _ffi.dt(w), R, K, int(group_size or 0), _ffi.ptr(out),
                                      _ffi.ptr(ws), _ffi.stream()), 'llmc_awq_weight_mean')
    return out


def awq_scales(x_mean, w_mean, ratio, version='v2'):
    """Awq.get_scales (awq.py:98-108)."""
    _ffi.require_gpu(x_mean, w_mean)
    L = _ffi.lib()
    out = torch.empty_like(x_mean)
    v = 1 if version == 'v1' else 2
    _ffi.check(L.llmc_awq_scales(_ffi.ptr(x_mean), _ffi.ptr(w_mean) if v == 1 else 0, _ffi.dt(x_mean), x_mean.numel(),
                                 float(ratio), v, _ffi.ptr(out), _ffi.stream()), 'llmc_awq_scales')
    return out


def scale_fakequant(w, scales, wquantizer):
    """fake_quant_weight_dynamic(w.mul_(scales)) (awq.py:147-164); returns a new tensor, w is untouched."""
    _ffi.require_gpu(w, scales)
    L = _ffi.lib()
    w = w.contiguous()
    R, K = w.shape
    g = wquantizer.group_size if wquantizer.granularity == 'per_group' else K
    out = torch.empty_like(w)
    _ffi.check(L.llmc_awq_scale_fakequant(_ffi.ptr(w), _ffi.ptr(scales.contiguous()), _ffi.dt(w), R, K, g,
                                          int(wquantizer.sym), float(wquantizer.qmin), float(wquantizer.qmax),
                                          _ffi.ptr(out), _ffi.stream()), 'llmc_awq_scale_fakequant')
    return out


def div_cols(x, scales, tiled=False):
    """scaling_input (base_blockwise_quantization.py:877-889): x / scales.view(1, -1).
    tiled: the quotient is written as a ktile_pack image (2-D, for linear_loss_sum(..., tiled=True))."""
    _ffi.require_gpu(x, scales)
    L = _ffi.lib()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    out = torch.empty_like(x2)
    fn, name = (L.llmc_div_cols_kt, 'llmc_div_cols_kt') if tiled else (L.llmc_div_cols, 'llmc_div_cols')
    _ffi.check(fn(_ffi.ptr(x2), _ffi.ptr(scales.contiguous()), _ffi.dt(x2), x2.shape[0], x2.shape[1], _ffi.ptr(out),
                  _ffi.stream()), name)
    return out if tiled else out.reshape(x.shape)


def mul_cols_(w, scales):
    """fc.weight.mul_(scales.view(1, -1)) (base_blockwise_quantization.py:770)."""
    _ffi.require_gpu(w, scales)
    L = _ffi.lib()
    assert w.is_contiguous()
    _ffi.check(L.llmc_mul_cols(_ffi.ptr(w), _ffi.ptr(scales.contiguous()), _ffi.dt(w), w.shape[0], w.shape[1],
                               _ffi.stream()), 'llmc_mul_cols')
    return w


def clamp_groups_(w, min_val, max_val, group_size):
    """apply_clip v1 (auto_clip.py:194-212)."""
    _ffi.require_gpu(w, min_val, max_val)
    L = _ffi.lib()
    assert w.is_contiguous()
    _ffi.check(L.llmc_clamp_groups(_ffi.ptr(w), _ffi.dt(w), w.shape[0], w.shape[1], int(group_size or 0),
                                   _ffi.ptr(min_val.contiguous()), _ffi.ptr(max_val.contiguous()), _ffi.stream()),
               'llmc_clamp_groups')
    return w


def linear_supported(x, w):
    """Shapes / dtypes llmc_linear_eval takes: 16-bit operands of one dtype, K % 64 == 0, every operand < 4 GiB."""
    if not (x.is_cuda and w.is_cuda and x.dtype == w.dtype and x.dtype in (torch.float16, torch.bfloat16) and w.dim() == 2):
        return False
    K, R = w.shape[1], w.shape[0]
    if x.shape[-1] != K or K % 64 != 0 or x.numel() == 0:
        return False
    N = x.numel() // K
    return max(N * K, R * K, N * R) * 2 < (1 << 32)


def linear_out(x, wq, bias=None, tiled=False, blocked=False):
    """F.linear(x, wq, bias) on the HIP GEMM: [N, K] x [R, K]^T (+ b) -> [N, R], rounded once to the model dtype.
    tiled: x and wq are ktile_pack images (x 2-D). blocked (tiled only): the result is the opaque tile-blocked image
    (LLMC_LINEAR_YBLOCKED) only linear_loss_sum(..., y0_blocked=True) and unblock_y read."""
    _ffi.require_gpu(x, wq, bias)
    L = _ffi.lib()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    wq = wq.contiguous()
    N, K = x2.shape
    R = wq.shape[0]
    if blocked:
        if not tiled:
            raise ValueError('linear_out: blocked output needs tiled operands')
        y = torch.empty(L.llmc_linear_eval_yblocked_bytes(N, R) // 2, dtype=x.dtype, device=x.device)
    else:
        y = torch.empty((N, R), dtype=x.dtype, device=x.device)
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    fn, name = (L.llmc_linear_eval_kt, 'llmc_linear_eval_kt') if tiled else (L.llmc_linear_eval, 'llmc_linear_eval')
    # the k-tiled kernel cuts a product that would leave CUs idle into k-slices when it is given the workspace for them
    # (llmc_linear_eval_ws_bytes covers it; products that fill the chip need none)
    ws = None
    if tiled and not blocked and L.llmc_linear_eval_ws_bytes(N, K, R) > 4 * ((N + 255) // 256) * ((R + 255) // 256):
        ws = _ffi.workspace(L.llmc_linear_eval_ws_bytes(N, K, R), x.device)
    _ffi.check(fn(_ffi.ptr(x2), _ffi.ptr(wq), _ffi.dt(x2), N, K, R, _ffi.LINEAR_YBLOCKED if blocked else 0, _ffi.ptr(y),
                  _ffi.ptr(bias), 0, _ffi.ptr(ws), _ffi.stream()), name)
    return y if blocked else y.reshape(*x.shape[:-1], R)


_FALLBACK_SEEN = set()


KTILE_MIN_TOKENS = 256      # below one 256-token tile the two k-tile packs (x and W) cost more than they save:
#                             such calls (per-token forwards of OriginFloatLinear / FakeQuantLinear) take the row-major kernel


def linear_auto(x, w, bias=None, xcache=None):
    """F.linear(x, w, bias) on the best HIP GEMM the shapes allow — same bits on both HIP routes:
      * k-tiled operands + the one-wave-per-SIMD kernel (llmc_linear_eval_kt) when K % 128 == 0 and there are enough
        rows to fill 256-row tiles: x and w are packed k-tiled first (one HBM pass each). `xcache` (a dict owned by the
        caller) keeps the packed image of the LAST activation tensor, so layers that consume the very same tensor
        (q / k / v, gate / up inside an inspected module) pack it once;
      * the row-major 8-wave kernel (llmc_linear_eval) for the remaining supported shapes;
      * anything else (K % 64 != 0, fp32 activations, operands >= 4 GiB) goes to the framework's GPU linear — said once
        per shape on stderr, never silently, never the CPU."""
    R, K = w.shape
    N = x.numel() // max(1, x.shape[-1])
    if N >= KTILE_MIN_TOKENS and ktile_supported(x, w) and _ffi.HOST_OPTIONS['awq_kt']:
        key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
        hit = xcache.get(key) if xcache is not None else None
        if hit is None:
            xt = ktile_pack(x)
            if xcache is not None:
                xcache.clear()
                xcache[key] = (x, xt)          # x is kept referenced: its storage cannot be re-used under the key
        else:
            xt = hit[1]
        y = linear_out(xt, ktile_pack(w), bias, tiled=True)
        return y.reshape(*x.shape[:-1], R)
    if linear_supported(x, w):
        return linear_out(x, w, bias)
    sig = (tuple(x.shape), tuple(w.shape), str(x.dtype), str(w.dtype))
    if sig not in _FALLBACK_SEEN:
        _FALLBACK_SEEN.add(sig)
        print(f'[llmc_amd] linear {sig}: outside the HIP GEMM\'s shape rules (16-bit operands of one dtype, K % 64 == 0, '
              'operands < 4 GiB) -> torch.nn.functional.linear on the GPU', file=sys.stderr)
    return torch.nn.functional.linear(x, w if w.dtype == x.dtype else w.to(x.dtype), bias)


def unblock_y(yb, N, R):
    """Tile-blocked image (linear_out(..., blocked=True)) -> row-major [N, R]. Index arithmetic in torch, for tests
    and debugging: tile (tm, tn), wave (wm, wn), accumulator (m, n), register r = 8 v + j, lane ->
    token tm*256 + wm*128 + m*32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column tn*256 + wn*128 + n*32 + (lane & 31)."""
    ntm, ntn = (N + 255) // 256, (R + 255) // 256
    t = yb.reshape(ntm, ntn, 2, 2, 4, 4, 2, 64, 8)            # tm tn wm wn m n v lane j
    t = t.reshape(ntm, ntn, 2, 2, 4, 4, 2, 2, 32, 2, 4)        # ... v lh l31 jh jl   (r = 8v + 4jh + jl)
    # token offset inside the 32-row accumulator block: (r & 3) + 8 (r >> 2) + 4 lh = jl + 8 (2v + jh) + 4 lh
    t = t.permute(0, 2, 4, 6, 9, 7, 10, 1, 3, 5, 8)            # tm wm m v jh lh jl | tn wn n l31
    return t.reshape(ntm * 256, ntn * 256)[:N, :R].contiguous()


def ktile_supported(x, w):
    """linear_supported and K % 128 == 0: the k-tiled GEMM (llmc_linear_eval_kt) takes the pair."""
    return linear_supported(x, w) and w.shape[1] % 128 == 0


def ktile_pack(m):
    """Row-major [rows, K] 16-bit matrix -> the k-tiled layout T[K/32][rows][32] llmc_linear_eval_kt streams (same bytes,
    a new buffer of the same shape: only linear_loss_sum(..., tiled=True) / linear_out(..., tiled=True) may read it)."""
    _ffi.require_gpu(m)
    m = m.reshape(-1, m.shape[-1]).contiguous()
    out = torch.empty_like(m)
    _ffi.check(_ffi.lib().llmc_ktile_pack(_ffi.ptr(m), _ffi.dt(m), m.shape[0], m.shape[1], _ffi.ptr(out), _ffi.stream()),
               'llmc_ktile_pack')
    return out


def linear_loss_sum(x, wq, y0, loss_acc=None, tiled=False, y0_blocked=False):
    """sum((y0 - F.linear(x, wq))^2) with the difference formed in the model dtype; returns / accumulates into
    a 1-element fp32 device tensor (no host sync). tiled: x and wq are ktile_pack images; y0_blocked: y0 is a
    linear_out(..., blocked=True) image."""
    _ffi.require_gpu(x, wq, y0)
    L = _ffi.lib()
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    wq = wq.contiguous()
    N, K = x2.shape
    R = wq.shape[0]
    if y0_blocked:
        if not tiled or y0.numel() * 2 != L.llmc_linear_eval_yblocked_bytes(N, R):
            raise ValueError('linear_loss_sum: y0_blocked needs tiled operands and a blocked image of the same [N, R]')
    else:
        y0 = y0.reshape(-1, y0.shape[-1]).contiguous()
    if loss_acc is None:
        loss_acc = torch.zeros(1, dtype=torch.float32, device=x.device)
    ws = _ffi.workspace(L.llmc_linear_eval_ws_bytes(N, K, R), x.device)
    fn, name = (L.llmc_linear_eval_kt, 'llmc_linear_eval_kt') if tiled else (L.llmc_linear_eval, 'llmc_linear_eval')
    _ffi.check(fn(_ffi.ptr(x2), _ffi.ptr(wq), _ffi.dt(x2), N, K, R, 1 | (_ffi.LINEAR_YBLOCKED if y0_blocked else 0), 0,
                  _ffi.ptr(y0), _ffi.ptr(loss_acc), _ffi.ptr(ws), _ffi.stream()), name)
    return loss_acc


def clip_search(w, x, wquantizer, clip_sym, n_grid=20, max_shrink=0.5):
    """AutoClipper.auto_clip_layer (auto_clip.py:84-191, v1, w_only). w [R,K]; x [n_tok,K] already subsampled.
    Returns (best_max [R, ng, 1], best_min [R, ng, 1]) in the model dtype."""
    _ffi.require_gpu(w, x)
    L = _ffi.lib()
    w, x = w.contiguous(), x.reshape(-1, x.shape[-1]).contiguous()
    R, K = w.shape
    g = wquantizer.group_size if wquantizer.granularity == 'per_group' else K
    ng = K // g
    bmax = torch.empty((R, ng, 1), dtype=w.dtype, device=w.device)
    bmin = torch.empty((R, ng, 1), dtype=w.dtype, device=w.device)
    _ffi.check(L.llmc_awq_clip_search(_ffi.ptr(w), _ffi.ptr(x), _ffi.dt(w), R, K, g, x.shape[0], int(n_grid),
                                      int(max_shrink * n_grid), int(bool(clip_sym)), int(wquantizer.sym),
                                      float(wquantizer.qmin), float(wquantizer.qmax), _ffi.ptr(bmax), _ffi.ptr(bmin),
                                      0, _ffi.stream()), 'llmc_awq_clip_search')
    return bmax, bmin


def clip_errs(w, x, wquantizer, clip_sym, n_grid=20, max_shrink=0.5):
    """The error table of clip_search: [n_shrink, R, ng] in the model dtype (auto_clip.py:150-180 `err` per shrink level)."""
    _ffi.require_gpu(w, x)
    L = _ffi.lib()
    w, x = w.contiguous(), x.reshape(-1, x.shape[-1]).contiguous()
    R, K = w.shape
    g = wquantizer.group_size if wquantizer.granularity == 'per_group' else K
    ns = int(max_shrink * n_grid)
    errs = torch.empty((ns, R, K // g), dtype=w.dtype, device=w.device)
    _ffi.check(L.llmc_awq_clip_errs(_ffi.ptr(w), _ffi.ptr(x), _ffi.dt(w), R, K, g, x.shape[0], int(n_grid), ns,
                                    int(bool(clip_sym)), int(wquantizer.sym), float(wquantizer.qmin),
                                    float(wquantizer.qmax), _ffi.ptr(errs), _ffi.stream()), 'llmc_awq_clip_errs')
    return errs


def clip_errs_cand(w, cands, x, xq, group_size):
    """The error table of auto_clip_layer from given candidates (llmc_awq_clip_errs_cand): w [R, K]; cands [ns, R, K] the
    fake-quantized weights of every shrink level; x [n_tok, K] the sampled tokens, xq their fake-quantized form (None:
    weight-only). Returns errs [ns, R, K / g] in the model dtype. The transposition of the token block ([K, n_tok], padded
    to 8 tokens) is the only data movement done here."""
    _ffi.require_gpu(w, cands, x, xq)
    L = _ffi.lib()
    w, cands = w.contiguous(), cands.contiguous()
    R, K = w.shape
    g = int(group_size or K)
    ns = cands.shape[0]
    x = x.reshape(-1, K)
    n_tok = x.shape[0]
    ldt = (n_tok + 7) // 8 * 8

    def transposed(t):
        out = torch.zeros((K, ldt), dtype=w.dtype, device=w.device)
        out[:, :n_tok] = t.reshape(-1, K).to(w.dtype).t()
        return out
    xt = transposed(x)
    xqt = xt if xq is None else transposed(xq)
    errs = torch.empty((ns, R, K // g), dtype=w.dtype, device=w.device)
    _ffi.check(L.llmc_awq_clip_errs_cand(_ffi.ptr(w), _ffi.ptr(cands), _ffi.ptr(xt), _ffi.ptr(xqt), _ffi.dt(w), R, K, g,
                                         n_tok, ldt, ns, _ffi.ptr(errs), _ffi.stream()), 'llmc_awq_clip_errs_cand')
    return errs
