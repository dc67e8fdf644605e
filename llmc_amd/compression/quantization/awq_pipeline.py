"""AWQ scale search for a set of Linear layers that share one input, inspect = the layers themselves.

Mirrors Awq.search_scale_subset (awq.py:179-278) for the shipped default of one calibration batch
(calib.bs = -1): 20-point grid over ratio, s = get_scales(...), Wq = fakequant(W * s), out = (x / s) Wq^T,
loss = mean((org_out - out)^2), argmin.  Differences in mechanics, not in arithmetic: nothing is copied to the
host and restored (the reference reloads the module's state dict from a CPU copy every grid step), the
activation mean is computed once instead of once per grid step, and the 20 losses stay on the device until
the final argmin (one sync instead of 20 `.item()` calls).
"""
import os

import torch

from llmc_amd import _ffi

from . import awq_ops


@torch.no_grad()
def search_scale_stacked(weights, x, wquantizer, trans_version='v2', n_grid=20, return_losses=False, timing=None):
    """weights: list of [R_i, K] tensors (model dtype, untouched); x: [..., K] activations (model dtype).
    Returns best_scales [K] (model dtype) (and the device tensor of the 20 mean losses)."""
    _ffi.require_gpu(x, *weights)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    N = x2.shape[0]
    wcat = torch.cat(weights, dim=0) if len(weights) > 1 else weights[0]
    R = wcat.shape[0]
    g = wquantizer.group_size if wquantizer.granularity == 'per_group' else 0
    # get_weight_scale (awq.py:48-72): per-layer mean, summed in the model dtype, divided by the layer count
    w_max = None
    for w in weights:
        m = awq_ops.weight_mean(w, g)
        w_max = m if w_max is None else w_max.add_(m)
    w_max = w_max.div_(len(weights))
    x_mean = awq_ops.act_mean(x2)                       # get_act_scale (awq.py:74-76)
    def _ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e
    # K % 128 == 0: the 21 products run on the k-tiled GEMM (llmc_linear_eval_kt), operands re-laid by llmc_ktile_pack.
    # The output rows are walked in chunks whose [N, R_c] reference output stays below the GEMM's 4-GiB offset range
    # (70B-class gate|up stacks at 65 536 tokens): the loss is a sum over outputs, so chunk sums add up.
    kt_ok = x2.is_cuda and awq_ops.ktile_supported(x2, wcat[:min(R, 256)]) and _ffi.HOST_OPTIONS['awq_kt']
    lim = int(_ffi.HOST_OPTIONS['awq_y_bytes'])      # test hook: smaller chunks
    rc = max(256, (lim // (2 * max(N, K))) // 256 * 256)
    chunks = [(r0, min(R, r0 + rc)) for r0 in range(0, R, rc)] if kt_ok else [(0, R)]
    kt = kt_ok and all(awq_ops.ktile_supported(x2, wcat[r0:r1]) for r0, r1 in chunks)
    if not kt:
        chunks = [(0, R)]
    xa = awq_ops.ktile_pack(x2) if kt else x2
    org_out = []
    for r0, r1 in chunks:                               # get_original_out (awq.py:128-132)
        wa = awq_ops.ktile_pack(wcat[r0:r1]) if kt else wcat[r0:r1]
        e0 = _ev() if timing is not None else None
        org_out.append(awq_ops.linear_out(xa, wa, tiled=kt, blocked=kt))
        if timing is not None:
            timing.append((e0, _ev(), 2.0 * N * (r1 - r0) * K))
    del xa, wa
    losses = torch.zeros(n_grid, dtype=torch.float32, device=x.device)
    scales_all = []
    for n in range(n_grid):
        ratio = n * 1 / n_grid
        s = awq_ops.awq_scales(x_mean, w_max, ratio, trans_version)
        xs = awq_ops.div_cols(x2, s, tiled=kt)
        for ci, (r0, r1) in enumerate(chunks):
            wq = awq_ops.scale_fakequant(wcat[r0:r1], s, wquantizer)
            if kt:
                wq = awq_ops.ktile_pack(wq)
            e0 = _ev() if timing is not None else None
            awq_ops.linear_loss_sum(xs, wq, org_out[ci], losses[n:n + 1], tiled=kt, y0_blocked=kt)
            if timing is not None:
                timing.append((e0, _ev(), 2.0 * N * (r1 - r0) * K))
        scales_all.append(s)
    losses /= float(N * R)                               # .pow(2).mean() (awq.py:136)
    best = int(torch.argmin(losses).item())              # strict '<' keeps the first minimum (awq.py:245)
    if return_losses:
        return scales_all[best], losses, best
    return scales_all[best]
