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
