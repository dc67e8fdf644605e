"""Numpy restatement of llmc's `calib_algo: static_hist` activation range (llmc/compression/quantization/quant.py):

  get_static_hist_range      quant.py:462-512   per-sample histograms (2048 bins) merged into the running range
  _combine_histograms        quant.py:368-401   re-binning of the old histogram when the range grows
  _upscale_histogram         quant.py:333-366   16x upsampling, mid-points bucketized into the new bins
  get_hist_threshold         quant.py:403-460   PyTorch's HistogramObserver search: shrink the range from both ends in
                                                1e-8 quantile steps while the L2 quantization error decreases
  get_quantization_error     quant.py:279-331   closed-form error of a uniform density per source bin
TEST INFRASTRUCTURE (oracle). Tensors are fp32 like the reference's (Python scalars enter an op at fp32, reductions are
torch's CPU reductions: the final `norm.sum()` is pinned by tolerance, everything discrete — bins, start / end — exactly,
tests/golden/hist.npz).
"""
import numpy as np

F = np.float32
BINS, UPS = 2048, 16


def histc(x, bins, lo, hi):
    """torch.histc on fp32: bin = int((x - lo) * bins / (hi - lo)), the right edge belongs to the last bin, values
    outside [lo, hi] are ignored; lo == hi (a single value) widens the range by one on both sides like ATen does."""
    x = np.asarray(x, F).reshape(-1)
    lo, hi = F(lo), F(hi)
    if lo == hi:
        lo, hi = F(lo - 1), F(hi + 1)
    ok = (x >= lo) & (x <= hi)
    pos = ((x[ok] - lo) * F(bins) / (hi - lo)).astype(np.int64)
    pos = np.minimum(pos, bins - 1)
    return np.bincount(pos, minlength=bins).astype(F)


def linspace(start, end, steps):
    """torch.linspace for fp32 on CPU: step = (end - start) / (steps - 1); the first half counts up from start, the
    second half down from end."""
    start, end = F(start), F(end)
    step = F((end - start) / F(steps - 1))
    i = np.arange(steps)
    half = steps // 2
    up = (start + step * i.astype(F)).astype(F)
    down = (end - step * (steps - 1 - i).astype(F)).astype(F)
    return np.where(i < half, up, down).astype(F)


def upscale_histogram(hist, orig_min, orig_max, update_min, update_max):
    h = (np.repeat(np.asarray(hist, F), UPS) / F(UPS)).astype(F)
    bin_size = F((F(orig_max) - F(orig_min)) / F(BINS * UPS))
    mid = (linspace(orig_min, orig_max, BINS * UPS + 1)[:-1] + F(0.5) * bin_size).astype(F)
    bounds = linspace(update_min, update_max, BINS + 1)
    bucket = np.searchsorted(bounds, mid, side='right') - 1        # bucketize(right=True) - 1
    bucket = np.clip(bucket, 0, BINS - 1)
    out = np.zeros(BINS, F)
    for b, w in zip(bucket, h):                                    # bincount(weights): sequential fp32 accumulation
        out[b] = F(out[b] + w)
    return out


def combine_histograms(orig_hist, orig_min, orig_max, update_hist, update_min, update_max):
    if update_min == orig_min and update_max == orig_max:
        return (orig_hist + update_hist).astype(F)
    if orig_min == orig_max:
        bin_value = F(np.sum(update_hist, dtype=F))
        return (histc(np.array([orig_min], F), BINS, update_min, update_max) * bin_value + update_hist).astype(F)
    assert update_min <= orig_min and update_max >= orig_max
    return (update_hist + upscale_histogram(orig_hist, orig_min, orig_max, update_min, update_max)).astype(F)


def quantization_error(hist, min_val, max_val, next_start_bin, next_end_bin, dst_nbins):
    bin_width = (float(max_val) - float(min_val)) / BINS                      # Python floats (`.item()`)
    dst_bin_width = bin_width * (next_end_bin - next_start_bin + 1) / dst_nbins
    if dst_bin_width == 0.0:
        return 0.0
    src_bin = np.arange(BINS)
    begin = ((src_bin - next_start_bin).astype(F) * F(bin_width)).astype(F)
    end = (begin + F(bin_width)).astype(F)
    dbw = F(dst_bin_width)
    d_begin = np.clip(np.floor(begin / dbw), 0, dst_nbins - 1).astype(F)
    d_begin_center = ((d_begin + F(0.5)) * dbw).astype(F)
    d_end = np.clip(np.floor(end / dbw), 0, dst_nbins - 1).astype(F)
    density = (np.asarray(hist, F) / F(bin_width)).astype(F)

    def norm3(b, e):                                                        # get_norm: density * (e^3 - b^3) / 3
        b, e = np.asarray(b, F), np.asarray(e, F)
        return (density * ((e * e * e - b * b * b) / F(3)).astype(F)).astype(F)

    half = dst_bin_width / 2
    norm = np.zeros(BINS, F)
    norm = (norm + norm3((begin - d_begin_center).astype(F), (np.ones(BINS, F) * F(half)).astype(F))).astype(F)
    norm = (norm + ((d_end - d_begin - F(1)).astype(F) * norm3(F(-half), F(half))).astype(F)).astype(F)
    d_end_center = (d_end * dbw + F(half)).astype(F)
    norm = (norm + norm3(F(-half), (end - d_end_center).astype(F))).astype(F)
    return float(np.sum(norm, dtype=F))


def hist_threshold(hist, min_val, max_val, dst_nbins):
    hist = np.asarray(hist, F)
    bin_width = F((F(max_val) - F(min_val)) / F(BINS))
    total = float(np.sum(hist, dtype=F))
    csum = np.cumsum(hist, dtype=F)
    stepsize, alpha, beta = 1e-8, 0.0, 1.0
    start_bin, end_bin, norm_min = 0, BINS - 1, float('inf')
    while alpha < beta:
        next_alpha, next_beta = alpha + stepsize, beta - stepsize
        left, right = start_bin, end_bin
        while left < end_bin and csum[left] < F(next_alpha * total):
            left += 1
        while right > start_bin and csum[right] > F(next_beta * total):
            right -= 1
        next_start, next_end = start_bin, end_bin
        if (left - start_bin) > (end_bin - right):
            next_start, alpha = left, next_alpha
        else:
            next_end, beta = right, next_beta
        if next_start == start_bin and next_end == end_bin:
            continue
        norm = quantization_error(hist, min_val, max_val, next_start, next_end, dst_nbins)
        if norm > norm_min:
            break
        norm_min, start_bin, end_bin = norm, next_start, next_end
    new_min = F(F(min_val) + bin_width * F(start_bin))
    new_max = F(F(min_val) + bin_width * F(end_bin + 1))
    return new_min, new_max, start_bin, end_bin


def static_hist_range(samples, dt, bit=8):
    """samples: list of arrays (values of the 16-bit dtype `dt`, fp32 container), one per calibration sample.
    Returns (new_min, new_max, histogram, min, max, start_bin, end_bin)."""
    hist, mn, mx = None, None, None
    for s in samples:
        s = np.asarray(s, F)
        x_min, x_max = F(s.min()), F(s.max())
        if hist is None:
            hist, mn, mx = histc(s, BINS, x_min, x_max), x_min, x_max
            continue
        new_min, new_max = min(mn, x_min), max(mx, x_max)
        upd = histc(s, BINS, new_min, new_max)
        if new_min == mn and new_max == mx:
            hist = (hist + upd).astype(F)
        else:
            hist = combine_histograms(hist, mn, mx, upd, new_min, new_max)
        mn, mx = new_min, new_max
    a, b, sb, eb = hist_threshold(hist, mn, mx, 2 ** bit)
    return a, b, hist, mn, mx, sb, eb
