"""Numpy restatement of llmc's GPTQ numerics (llmc/compression/quantization/gptq.py).

  add_batch                      gptq.py:254-295   running-mean Hessian H = (2/n) sum_b X_b^T X_b
  hessian_sorting                gptq.py:58-64     perm = argsort(diag(H), descending)
  process_hessian_and_weights    gptq.py:128-176   dead columns, permute, damp, chol -> chol_inverse -> chol(upper)
  weight_transform               gptq.py:199-244   blocked column loop (quant, error feedback, trailing update)
  search_column_qparams          gptq.py:359-366   per-group min/max qparams from the CURRENT weights
Test infrastructure only (see oracle/__init__.py). fp32 throughout, like the reference; matmul / Cholesky
summation order is the platform BLAS's, so H and U are pinned by tolerance, the column loop bit-exactly
(given identical W and U) — tests/test_oracle_golden.py.
"""
import math

import numpy as np

from . import quant_ref as Q


def add_batch(H, nsamples, inp, chunk_num=1):
    """gptq.py:254-295 for one hook call. inp: [b, seq, K] or [seq, K] (values of a 16-bit dtype, fp32
    container). Returns (H, nsamples). H fp32 [K, K] is updated in place."""
    inp = np.asarray(inp, dtype=np.float32)
    if inp.ndim == 2:
        inp = inp[None]
    tmp = inp.shape[0]
    x = inp.reshape(-1, inp.shape[-1]).T  # [K, tokens]
    H *= np.float32(nsamples / (nsamples + tmp))
    nsamples += tmp
    scale = np.float32(math.sqrt(2 / nsamples))
    for chunk in np.array_split(x, chunk_num, axis=1):
        c = (scale * chunk).astype(np.float32)
        H += c @ c.T
    return H, nsamples


def hessian_exact(batches):
    """fp64 ground truth of what add_batch converges to: (2 / n_batches) * sum X^T X."""
    n = 0
    acc = None
    for b in batches:
        b = np.asarray(b, dtype=np.float64)
        if b.ndim == 2:
            b = b[None]
        n += b.shape[0]
        x = b.reshape(-1, b.shape[-1])
        g = x.T @ x
        acc = g if acc is None else acc + g
    return acc * (2.0 / n)


# ---- K2/K3: prep + factorisation (tolerance-pinned: LAPACK order differs from the reference's MKL) ----
def hessian_sorting(H):
    """gptq.py:63 — argsort(diag(H), descending). Ties (dead columns) are interchangeable."""
    return np.argsort(-np.diag(H), kind='stable')


def process_hessian_and_weights(W, H, perm=None, percdamp=0.01):
    """gptq.py:135-174 -> (W' fp32 [R,K] permuted/dead-fixed, U fp32 upper with H^-1 = U^T U)."""
    import scipy.linalg
    W = np.array(W, dtype=np.float32, copy=True)
    H = np.array(H, dtype=np.float32, copy=True)
    K = H.shape[0]
    dead = np.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    if perm is not None:
        W = W[:, perm]
        H = H[perm][:, perm]
    damp = np.float32(percdamp) * np.mean(np.diag(H), dtype=np.float32)
    H[np.arange(K), np.arange(K)] += damp
    L = np.linalg.cholesky(H.astype(np.float32))
    Linv = scipy.linalg.solve_triangular(L, np.eye(K, dtype=np.float32), lower=True)
    Hinv = (Linv.T @ Linv).astype(np.float32)
    U = np.linalg.cholesky(Hinv).T.astype(np.float32)
    return W, np.ascontiguousarray(U)


# ---- K4: the column loop, C restatement (oracle/csrc/gptq_canon.c) ---------------------------------
_lib = None


def _clib():
    global _lib
    if _lib is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, '_build', 'liboracle_gptq.so')
        src = os.path.join(here, 'csrc', 'gptq_canon.c')
        src2 = os.path.join(here, 'csrc', 'spqr_canon.c')
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(src2)):
            subprocess.check_call(['make', '-C', here, '-s'])
        _lib = ctypes.CDLL(so)
        _lib.gptq_weight_transform.restype = ctypes.c_int
    return _lib


def owq_permutation(h_diag, n_out):
    """hessian_sorting with OWQ (gptq.py:66-83, actorder forced off): non-outlier columns in original order, then
    the n_out largest Hessian diagonals, largest first (stable ties like torch.argsort's default are not relied on)."""
    desc = np.argsort(-np.asarray(h_diag, dtype=np.float32), kind='stable')
    keep = np.ones(len(h_diag), dtype=bool)
    keep[desc[:n_out]] = False
    return np.concatenate([np.arange(len(h_diag))[keep], desc[:n_out]]).astype(np.int64)


def weight_transform(W, Hinv, sym, qmin, qmax, group_size, static_groups=False, col_group=None,
                     scales=None, zeros=None, blocksize=128, want_losses=True, n_quant=None, init_scales=None,
                     init_zeros=None):
    """gptq.py:199-244. W [R,K] fp32 (copied), Hinv [K,K] fp32 upper.
    Returns dict(tmp, W (running), losses, scales [R,ng], zeros [R,ng]).
    n_quant < K: OWQ (only the first n_quant columns are visited; dynamic qparams of never-visited groups keep
    init_scales / init_zeros, the layer's RTN qparams the reference's `self.groups` starts from)."""
    import ctypes
    L = _clib()
    W = np.array(W, dtype=np.float32, copy=True, order='C')
    Hinv = np.ascontiguousarray(Hinv, dtype=np.float32)
    R, K = W.shape
    per_channel = not group_size
    ng = 1 if per_channel else -(-K // group_size)
    static_mode = static_groups or per_channel
    if static_mode:
        scales = np.ascontiguousarray(np.asarray(scales, dtype=np.float32).reshape(R, ng))
        zeros = None if zeros is None else np.ascontiguousarray(np.asarray(zeros, dtype=np.float32).reshape(R, ng))
    else:
        scales = (np.array(init_scales, dtype=np.float32).reshape(R, ng).copy() if init_scales is not None
                  else np.zeros((R, ng), dtype=np.float32))
        zeros = (np.array(init_zeros, dtype=np.float32).reshape(R, ng).copy() if init_zeros is not None
                 else np.zeros((R, ng), dtype=np.float32))
    cg = None if col_group is None else np.ascontiguousarray(col_group, dtype=np.int32)
    tmp = np.zeros_like(W)
    losses = np.zeros_like(W) if want_losses else None

    def p(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    rc = L.gptq_weight_transform_cols(p(W), p(Hinv), ctypes.c_int64(R), ctypes.c_int64(K),
                                      ctypes.c_int64(K if n_quant is None else int(n_quant)), int(bool(sym)),
                                      ctypes.c_float(qmin), ctypes.c_float(qmax), ctypes.c_int64(group_size or 0),
                                      int(bool(static_groups)), p(cg), p(scales), p(zeros), p(tmp), p(losses),
                                      int(blocksize))
    assert rc == 0
    return dict(tmp=tmp, W=W, losses=losses, scales=scales, zeros=zeros)


def mm_chain(a, b):
    import ctypes
    L = _clib()
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((a.shape[0], b.shape[1]), dtype=np.float32)
    L.mm_chain(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
               out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(a.shape[0]), ctypes.c_int64(a.shape[1]),
               ctypes.c_int64(b.shape[1]))
    return out
