"""Numpy restatement of llmc's AWQ scale search (llmc/compression/quantization/awq.py) and of the pieces of
BaseBlockwiseQuantization it calls.

  get_weight_scale        awq.py:48-72     mean over rows (and layers) of |W| / group-max(|W|)
  get_act_scale           awq.py:74-85     mean over tokens of |x|
  get_scales              awq.py:88-108    x_mean^ratio (v2) or x^r / w^(1-r) (v1), clamp 1e-4, / sqrt(max*min)
  fake_quantize_weight    awq.py:147-164   fakequant_dyn(W * s) in the model dtype
  scaling_input           base_blockwise_quantization.py:877-889   x / s
  calculate_loss          awq.py:134-136   mean((org_out - out).float()^2)
  search_scale_subset     awq.py:179-253   20-point grid, one calibration batch, argmin
Test infrastructure only (see oracle/__init__.py). Reductions (mean over 10^4..10^5 tokens, the GEMM's k-sum)
are fp32 sums whose order is the BLAS/ATen implementation's; they are pinned by tolerance (scales usually
bit-equal after the rounding to 16 bit), the elementwise chains bit-exactly.
"""
import numpy as np

from . import aten_sum

from . import quant_ref as Q
from .quant_ref import rnd


def act_mean(x, dt):
    x = np.asarray(x, dtype=np.float32).reshape(-1, x.shape[-1])
    s = np.abs(x).sum(axis=0, dtype=np.float32)
    return rnd(s / np.float32(x.shape[0]), dt)


def weight_scale(weights, dt, group_size):
    total = None
    for w in weights:
        w = np.asarray(w, dtype=np.float32)
        g = group_size or w.shape[1]
        a = np.abs(w.reshape(-1, g))
        m = a.max(axis=1, keepdims=True)
        with np.errstate(invalid='ignore', divide='ignore'):
            ls = rnd(a / m, dt).reshape(w.shape)
        mean = rnd(ls.sum(axis=0, dtype=np.float32) / np.float32(w.shape[0]), dt)
        total = mean if total is None else rnd(total + mean, dt)
    return rnd(total / np.float32(len(weights)), dt)


def get_scales(x_mean, w_mean, ratio, dt, version='v2'):
    # Tensor.pow(python_float) on CPU casts the exponent to the tensor dtype first (0.6 -> 0.60009765625 in f16)
    r = rnd(np.float32(ratio), dt)
    with np.errstate(invalid='ignore', divide='ignore'):
        if version == 'v1':
            a = rnd(np.power(x_mean, r, dtype=np.float32), dt)
            b = rnd(np.power(w_mean, rnd(np.float32(1.0 - ratio), dt), dtype=np.float32), dt)
            s = rnd(a / b, dt)
        else:
            s = rnd(np.power(x_mean, r, dtype=np.float32), dt)
        s = np.maximum(s, rnd(np.float32(1e-4), dt))
        den = rnd(np.sqrt(rnd(s.max() * s.min(), dt)), dt)
        return rnd(s / den, dt)


def fake_quantize_weight(w, s, dt, sym, qmin, qmax, group_size):
    ws = rnd(np.asarray(w, dtype=np.float32) * s[None, :], dt)
    g = group_size or w.shape[1]
    fq, _, _ = Q.fake_quant_dynamic(ws.reshape(-1, g), dt, sym, qmin, qmax)
    return fq.reshape(w.shape)


def scaling_input(x, s, dt):
    return rnd(np.asarray(x, dtype=np.float32) / s, dt)


def linear(x, w, dt):
    x2 = np.asarray(x, dtype=np.float32).reshape(-1, x.shape[-1])
    return rnd(x2 @ np.asarray(w, dtype=np.float32).T, dt)


def loss_mean(y0, y, dt):
    d = rnd(y0 - y, dt)
    return float(np.mean(d.astype(np.float32) ** 2, dtype=np.float32))


def search_scale(weights, x, dt, sym, qmin, qmax, group_size, version='v2', n_grid=20):
    """One calibration batch (the shipped default bs=-1), inspect = the Linear layers themselves (outputs
    concatenated). Returns (best_scales [K], losses [n_grid], best_n)."""
    x = np.asarray(x, dtype=np.float32)
    wcat = np.concatenate([np.asarray(w, dtype=np.float32) for w in weights], axis=0)
    w_max = weight_scale(weights, dt, group_size)
    y0 = linear(x, wcat, dt)
    x_mean = act_mean(x, dt)
    best, best_s, best_n = float('inf'), None, -1
    losses = []
    for n in range(n_grid):
        ratio = n * 1 / n_grid
        s = get_scales(x_mean, w_max, ratio, dt, version)
        wq = np.concatenate([fake_quantize_weight(w, s, dt, sym, qmin, qmax, group_size) for w in weights], axis=0)
        y = linear(scaling_input(x, s, dt), wq, dt)
        ls = loss_mean(y0, y, dt)
        losses.append(ls)
        if ls < best:
            best, best_s, best_n = ls, s, n
    return best_s, np.array(losses, dtype=np.float64), best_n


# ---- AutoClipper (llmc/compression/quantization/auto_clip.py:84-191, clip_version v1, w_only) ------------
def auto_clip_layer(w, x, dt, sym, qmin, qmax, group_size, clip_sym=True, n_grid=20, max_shrink=0.5,
                    n_sample_token=512, errs_out=None):
    """w [R,K], x [tokens,K] (values of dtype dt) or a LIST of such batches (auto_clip.py:130-184: the error is averaged
    over the list in dt). Returns (best_max [R, ng, 1], best_min [R, ng, 1]).
    Per (row, group), for i_s in range(int(max_shrink * n_grid)):
        max = org_max * (1 - i_s / n_grid) ; min = -max (clip_sym) | org_min * (1 - i_s / n_grid)
        q_w = fakequant_dyn(clamp(w, min, max)) ; err = mean_tok((sum_k x*q_w - sum_k x*w)^2)   -> argmin.
    All arithmetic in dt; `x * w` is rounded per product, the k-sum and the token mean accumulate in fp32 (ATen casts
    16-bit means to fp32, sums, divides and rounds once). Pinned 100 % to tests/golden/clip.npz and clip_mb.npz."""
    w = np.asarray(w, dtype=np.float32)
    R, K = w.shape
    g = group_size or K
    ng = K // g
    xs = x if isinstance(x, (list, tuple)) else [x]
    xgs = []
    for xi in xs:
        xi = np.asarray(xi, dtype=np.float32).reshape(-1, K)
        step = max(1, xi.shape[0] // n_sample_token)
        xi = xi[0::step]                              # auto_clip.py:146-147
        xgs.append(xi.reshape(xi.shape[0], ng, g))    # [tok, ng, g]
    wg = w.reshape(R, ng, g)
    org_max = np.abs(wg).max(axis=-1, keepdims=True) if clip_sym else wg.max(axis=-1, keepdims=True)
    org_min = wg.min(axis=-1, keepdims=True)

    def out_of(xg, wq):   # (x * w).sum(-1): [R, tok, ng]
        o = np.empty((R, xg.shape[0], ng), dtype=np.float32)
        for r in range(R):
            prod = rnd(xg * wq[r][None], dt)         # [tok, ng, g]
            # the k-sum in ATen's order (vectorized_inner_sum of a 16-bit tensor); shorter groups: plain fp32 sum
            ks = aten_sum.inner_sum_16bit(prod) if g >= 16 else prod.sum(axis=-1, dtype=np.float32)
            o[r] = rnd(ks, dt)
        return o

    org_outs = [out_of(xg, wg) for xg in xgs]
    best_max, best_min = org_max.copy(), org_min.copy()
    with np.errstate(over='ignore'):
        min_errs = rnd(np.full_like(org_max, 1e9), dt)
    for i_s in range(int(max_shrink * n_grid)):
        f = np.float32(1 - i_s / n_grid)             # Tensor * python_scalar keeps the scalar in fp32 opmath (ATen CPU mul)
        max_val = rnd(org_max * f, dt)
        min_val = -max_val if clip_sym else rnd(org_min * f, dt)
        cur_w = np.minimum(np.maximum(wg, min_val), max_val)
        qw, _, _ = Q.fake_quant_dynamic(cur_w.reshape(-1, g), dt, sym, qmin, qmax)
        err_mean = None
        for xg, org_out in zip(xgs, org_outs):
            cur_out = out_of(xg, qw.reshape(R, ng, g))
            d = rnd(cur_out - org_out, dt)
            sq = rnd(d * d, dt)
            # 16-bit mean: cast to fp32, sum over tokens in the serial iterator's order, divide, one rounding
            err = rnd(aten_sum.outer_sum_fp32(sq) / np.float32(sq.shape[1]), dt).reshape(R, ng, 1)
            err_mean = err if err_mean is None else rnd(err_mean + err, dt)      # err_mean = 0; err_mean += err
        err_mean = rnd(err_mean / np.float32(len(xgs)), dt)                      # err_mean /= len(inputs)
        if errs_out is not None:
            errs_out.append(err_mean.reshape(R, ng).copy())
        better = err_mean < min_errs
        min_errs = np.where(better, err_mean, min_errs)
        best_max = np.where(better, max_val, best_max)
        best_min = np.where(better, min_val, best_min)
    return best_max, best_min


# ---- AutoClipper from given candidates (auto_clip.py:150-184 with any quantizer / granularity / clip version) ---------
def clip_errs_from_candidates(w, cands, x, xq, dt, g):
    """The error table of auto_clip_layer when the candidates are given: w [R, K]; cands [ns, R, K] = the fake-quantized
    weights of every shrink level (v1: fakequant(clamp(w)), v2: static fake-quant with the learnable range); x [tok, K] the
    sampled tokens, xq their fake-quantized form (x itself when activations are not quantized, auto_clip.py:276-281).
    errs[s, r, j] = mean_tok(((xq * cands[s]).sum_g - (x * w).sum_g)^2), every op in dt, sums in ATen's CPU orders
    (oracle/aten_sum.py). What llmc_awq_clip_errs_cand computes."""
    w = np.asarray(w, dtype=np.float32)
    R, K = w.shape
    ng = K // g
    xg = np.asarray(x, dtype=np.float32).reshape(-1, ng, g)
    xqg = np.asarray(xq, dtype=np.float32).reshape(-1, ng, g)

    def out_of(xv, wv):
        o = np.empty((R, xv.shape[0], ng), dtype=np.float32)
        for r in range(R):
            o[r] = rnd(aten_sum.inner_sum_16bit(rnd(xv * wv[r][None], dt)), dt)
        return o

    org = out_of(xg, w.reshape(R, ng, g))
    errs = []
    for s in range(len(cands)):
        cur = out_of(xqg, np.asarray(cands[s], dtype=np.float32).reshape(R, ng, g))
        d = rnd(cur - org, dt)
        sq = rnd(d * d, dt)
        errs.append(rnd(aten_sum.outer_sum_fp32(sq) / np.float32(sq.shape[1]), dt))
    return np.stack(errs)                                   # [ns, R, ng]


def clip_argmin_levels(errs, w, g, dt, clip_sym, n_grid=20):
    """auto_clip.py:126-127, 152-158, 176-184: strict-< argmin over the shrink levels -> (best_max, best_min) [R, ng, 1]."""
    w = np.asarray(w, dtype=np.float32)
    R, K = w.shape
    wg = w.reshape(R, K // g, g)
    org_max = np.abs(wg).max(axis=-1, keepdims=True) if clip_sym else wg.max(axis=-1, keepdims=True)
    org_min = wg.min(axis=-1, keepdims=True)
    best_max, best_min = org_max.copy(), org_min.copy()
    with np.errstate(over='ignore'):
        min_errs = rnd(np.full_like(org_max, 1e9), dt)
    for i_s in range(errs.shape[0]):
        f = np.float32(1 - i_s / n_grid)
        max_val = rnd(org_max * f, dt)
        min_val = -max_val if clip_sym else rnd(org_min * f, dt)
        e = errs[i_s][..., None]
        better = e < min_errs
        min_errs = np.where(better, e, min_errs)
        best_max = np.where(better, max_val, best_max)
        best_min = np.where(better, min_val, best_min)
    return best_max, best_min


# ---- AWQ with activation quantization / other weight quantizers (awq.py:147-177, 223-224) ---------------------------------
def fake_quant_any(x, dt, kind, bit, sym, granularity, group_size=0):
    """fake_quant_{weight,act}_dynamic of IntegerQuantizer / FloatQuantizer (use_qtorch) on an array of dt values: ranges over
    the last dimension (per_channel / per_token), over groups of it (per_group) or over the whole array (per_tensor)."""
    x = np.asarray(x, dtype=np.float32)
    rows = Q.reshape_rows(x, 'per_channel' if granularity == 'per_token' else granularity, group_size or None)
    if kind == 'int':
        qmin, qmax = Q.int_range(int(bit), bool(sym))
        if granularity == 'per_tensor' and not sym:
            return Q.per_tensor_asym_fake_and_codes(x, dt, qmin, qmax)[0].reshape(x.shape)
        return Q.fake_quant_dynamic(rows, dt, bool(sym), qmin, qmax)[0].reshape(x.shape)
    return Q.fp8_fake(rows, dt, str(bit), 'qtorch').reshape(x.shape)


def wa_chain_point(w, x, s, dt, wcfg, acfg, per_sample=False):
    """One grid point of the W-A search: fake_quantize_weight = fake-quant of w * s (awq.py:155-156), fake_quantize_input =
    fake-quant of x / s, the whole batch or sample by sample (awq.py:166-177). wcfg / acfg = (kind, bit, sym, granularity
    [, group_size])."""
    wq = fake_quant_any(rnd(np.asarray(w, dtype=np.float32) * np.asarray(s, dtype=np.float32)[None, :], dt), dt, *wcfg)
    xs = scaling_input(x, s, dt)
    if per_sample:
        xq = np.stack([fake_quant_any(xs[i], dt, *acfg) for i in range(xs.shape[0])])
    else:
        xq = fake_quant_any(xs, dt, *acfg)
    return wq, xq


def _logit(x, dt):
    """AutoClipper.logit = log(x / (1 - x)) on a dt tensor: every op rounds to dt (auto_clip.py:41)."""
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        one_minus = rnd(np.float32(1.0) - x, dt)
        return rnd(np.log(rnd(x / one_minus, dt)).astype(np.float32), dt)


def _sigmoid(x, dt):
    with np.errstate(over='ignore'):
        return rnd((np.float32(1.0) / (np.float32(1.0) + np.exp(-x.astype(np.float32)))).astype(np.float32), dt)


def auto_clip_layer_general(w, x, dt, wcfg, acfg, clip_version='v1', clip_sym=True, n_grid=20, max_shrink=0.5, n_sample_token=512):
    """auto_clip_layer (auto_clip.py:84-191) for every quantizer kind, candidates included: per output-channel batch of 256 / 64 rows
    (auto_clip.py:106-107; a per_tensor range spans the batch) and shrink level, fake_quantize_weight (:258-274) =
      v1: fake_quant_dynamic(clamp(w, min, max));
      v2: static fake-quant with the learnable range (quant.py:205-224): min / max (or +-absmax) scaled by sigmoid(logit(level ratio))
    then the error table and the argmin above. wcfg = (kind, bit, sym, granularity[, group_size]); acfg likewise or None (weight-only).
    Integer weight quantizers for v2 (the only kind the reference's learnable range is used with)."""
    w = np.asarray(w, dtype=np.float32)
    R, K = w.shape
    kind, bit, sym, gran = wcfg[:4]
    g = int(wcfg[4]) if (gran == 'per_group' and len(wcfg) > 4 and wcfg[4]) else K
    ng = K // g
    xs = np.asarray(x, dtype=np.float32).reshape(-1, K)
    xs = xs[0::max(1, xs.shape[0] // n_sample_token)]
    xq = xs if acfg is None else fake_quant_any(xs.reshape(1, xs.shape[0], ng, g), dt, *acfg).reshape(xs.shape)
    wg = w.reshape(R, ng, g)
    org_max = np.abs(wg).max(axis=-1, keepdims=True) if clip_sym else wg.max(axis=-1, keepdims=True)
    org_min = wg.min(axis=-1, keepdims=True)
    oc = 256 if R % 256 == 0 else 64
    cands = []
    for i_s in range(int(max_shrink * n_grid)):
        f = np.float32(1 - i_s / n_grid)
        max_val = rnd(org_max * f, dt)
        min_val = -max_val if clip_sym else rnd(org_min * f, dt)
        rows = []
        for b0 in range(0, R, oc):
            sl = slice(b0, b0 + oc)
            if clip_version == 'v1':
                cur = np.minimum(np.maximum(wg[sl], min_val[sl]), max_val[sl]).reshape(oc, K)
                rows.append(fake_quant_any(cur, dt, *wcfg))
            else:
                assert kind == 'int' and gran != 'per_group'
                qmin, qmax = Q.int_range(int(bit), bool(sym))
                with np.errstate(divide='ignore', invalid='ignore'):
                    low = _logit(rnd(min_val[sl] / org_min[sl], dt), dt)
                    up = _logit(rnd(max_val[sl] / org_max[sl], dt), dt)
                mn, mx = wg[sl].min(axis=-1, keepdims=True), wg[sl].max(axis=-1, keepdims=True)
                if sym:
                    a = np.maximum(np.maximum(np.abs(mx), np.abs(mn)), rnd(np.float32(1e-5), dt))
                    a = rnd(_sigmoid(up, dt) * a, dt)
                    mn, mx = -a, a
                else:
                    mn, mx = rnd(_sigmoid(low, dt) * mn, dt), rnd(_sigmoid(up, dt) * mx, dt)
                s, z = Q.qparams_from_minmax(mn.reshape(-1, 1), mx.reshape(-1, 1), dt, bool(sym), qmin, qmax)
                rows.append(Q.fake_quant_static(wg[sl].reshape(-1, g), dt, s, dt, z, dt, qmin, qmax).reshape(oc, K))
        cands.append(np.concatenate(rows, axis=0))
    errs = clip_errs_from_candidates(w, np.stack(cands), xs, xq, dt, g)
    return clip_argmin_levels(errs, w, g, dt, clip_sym, n_grid)


def search_scale_wa(w, xs, dt, wcfg, acfg, version='v2', n_grid=20):
    """Awq.search_scale_subset with activation quantization for ONE layer that is its own inspected module and a list of
    calibration batches (awq.py:179-253): per grid point and batch, loss = mean((x W^T - fq(x / s) fq(W s)^T)^2); the per-batch
    bookkeeping of the reference (loss_mean += b / n_samples * loss, best taken inside the batch loop). Returns (best_scales,
    losses [n_grid * len(xs)])."""
    w = np.asarray(w, dtype=np.float32)
    gsz = int(wcfg[4]) if (wcfg[3] == 'per_group' and len(wcfg) > 4) else 0
    w_max = weight_scale([w], dt, gsz)
    xs = [np.asarray(x, dtype=np.float32) for x in xs]
    n_samples = sum(x.shape[0] for x in xs)
    y0 = [linear(x, w, dt).reshape(*x.shape[:-1], -1) for x in xs]
    best, best_s, losses = float('inf'), None, []
    for n in range(n_grid):
        loss_mean, scales_mean = 0.0, 0.0
        for i, x in enumerate(xs):
            s = get_scales(act_mean(x, dt), w_max, n / n_grid, dt, version)
            wq, xq = wa_chain_point(w, x, s, dt, wcfg, acfg)
            ls = loss_mean_fn(y0[i], linear(xq, wq, dt).reshape(y0[i].shape), dt)
            losses.append(ls)
            frac = x.shape[0] / (x.shape[0] if len(xs) == 1 else n_samples)
            loss_mean += frac * ls
            scales_mean = scales_mean + frac * s
            if loss_mean < best:
                best, best_s = loss_mean, scales_mean
    return best_s, np.array(losses, dtype=np.float64)


def loss_mean_fn(y0, y, dt):
    return loss_mean(y0, y, dt)


# ---- block-wise FP8 checkpoints (`weight` float8_e4m3fn + `weight_scale_inv` per b x b block; awq.py:53-58, 147-164,
# base_blockwise_quantization.py:46-68, 655-700, 750-775). The reference de-blocks to bf16 (quant.py:18-30: fp32 product of
# code value and block scale, rounded to bf16 once), works in bf16, and re-blocks (quant.py:33-43 = FloatQuantizer e4m3
# per_block). `sem`: 'qtorch' = the reference's FloatQuantizer spelling, 'cast' = its Triton kernels' e4m3fn cast.
def fp8ckpt_to_bf16(bits, scales, block):
    M, N = bits.shape
    s = np.repeat(np.repeat(np.asarray(scales, np.float32), block, 0), block, 1)[:M, :N]
    return rnd((Q.e4m3fn_bits_to_f32(bits) * s).astype(np.float32), 'bf16')


def fp8ckpt_from_bf16(w, block, sem='qtorch'):
    bits, scales, _ = Q.fp8_per_block(w, 'bf16', block, sem)
    return bits, scales


def fp8ckpt_weight_scale(layers, block, group_size):
    """get_weight_scale over FP8 layers [(bits, scales), ...]"""
    return weight_scale([fp8ckpt_to_bf16(b, s, block) for b, s in layers], 'bf16', group_size)


def fp8ckpt_fake_quantize_weight(bits, scales, cols, block, sym, qmin, qmax, group_size, sem='qtorch'):
    w = fp8ckpt_to_bf16(bits, scales, block)
    return fp8ckpt_from_bf16(fake_quantize_weight(w, cols, 'bf16', sym, qmin, qmax, group_size), block, sem)


def fp8ckpt_w_qdq(bits, scales, block, sym, qmin, qmax, group_size, sem='qtorch'):
    w = fp8ckpt_to_bf16(bits, scales, block)
    g = group_size or w.shape[1]
    fq, _, _ = Q.fake_quant_dynamic(w.reshape(-1, g), 'bf16', sym, qmin, qmax)
    return fp8ckpt_from_bf16(fq.reshape(w.shape), block, sem)


def fp8ckpt_mul_cols(bits, scales, cols, block, sem='qtorch'):
    """fc.weight.mul_(scales.view(1, -1)) of scale_ln_fcs / scale_fc_fc on an FP8 layer"""
    w = rnd(fp8ckpt_to_bf16(bits, scales, block) * np.asarray(cols, np.float32)[None, :], 'bf16')
    return fp8ckpt_from_bf16(w, block, sem)


def fp8ckpt_div_rows(bits, scales, rows, block, sem='qtorch'):
    """fc1.weight.div_(scales.view(-1, 1)) of scale_fc_fc on an FP8 layer"""
    w = rnd(fp8ckpt_to_bf16(bits, scales, block) / np.asarray(rows, np.float32)[:, None], 'bf16')
    return fp8ckpt_from_bf16(w, block, sem)
