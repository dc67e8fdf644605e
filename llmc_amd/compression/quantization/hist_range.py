"""`calib_algo: static_hist` (llmc/compression/quantization/quant.py:264-512): the activation range PyTorch's histogram
observer would pick, as the reference's IntegerQuantizer computes it for per-tensor symmetric static activation quantization.

The pass over the data — per-sample min / max and torch.histc over the running range — runs on the GPU (`llmc_histc`,
act_hist.hip). What follows works on 2048 numbers and is host control logic, like in the reference (which synchronises
with `.item()` in every step of the search): merging a sample's histogram into the running one (re-binning the old one
through a 16x upsampled copy when the range grows) and shrinking the range from both ends in 1e-8 quantile steps while
the closed-form L2 quantization error decreases. fp32 where the reference's tensors are fp32, Python floats where it
uses `.item()`."""
import numpy as np
import torch

from llmc_amd import _ffi

F = np.float32


def sample_minmax(samples):
    """fp32 (min, max) of every calibration sample, [n] each on the device: `sample.min()`, `sample.max()` of quant.py:253-263,
    524-543 for all samples in one launch pair per 160 samples (`llmc_minmax_samples`: the samples are separate allocations,
    their addresses travel in the kernel arguments). The values are elements of the samples, so fp32 holds them exactly."""
    import ctypes
    _ffi.require_gpu(*samples)
    L = _ffi.lib()
    xs = [s if s.is_contiguous() else s.contiguous() for s in samples]
    assert all(x.dtype == xs[0].dtype for x in xs), 'calibration samples of one input share a dtype'
    dev = xs[0].device
    mn = torch.empty(len(xs), dtype=torch.float32, device=dev)
    mx = torch.empty(len(xs), dtype=torch.float32, device=dev)
    cap = L.llmc_minmax_samples_max()
    for i0 in range(0, len(xs), cap):
        part = xs[i0:i0 + cap]
        n = len(part)
        ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in part])
        lens = (ctypes.c_int64 * n)(*[x.numel() for x in part])
        ws = _ffi.workspace(L.llmc_minmax_samples_ws_bytes(lens, n), dev)
        _ffi.check(L.llmc_minmax_samples(ptrs, lens, n, _ffi.dt(part[0]), mn[i0:].data_ptr(), mx[i0:].data_ptr(), _ffi.ptr(ws),
                                         _ffi.stream()), 'llmc_minmax_samples')
    return mn, mx


class HistRange:
    def __init__(self, bins=2048, upsample_rate=16, dst_nbins=256):
        self.bins, self.ups, self.dst = int(bins), int(upsample_rate), int(dst_nbins)
        self.hist = None
        self.lo = self.hi = None

    # ---- data pass (GPU) ---------------------------------------------------------------------------------------------
    def _histc(self, x, lo, hi):
        L = _ffi.lib()
        out = torch.empty(self.bins, dtype=torch.float32, device=x.device)
        ws = _ffi.workspace(L.llmc_histc_ws_bytes(self.bins), x.device)
        _ffi.check(L.llmc_histc(_ffi.ptr(x), _ffi.dt(x), x.numel(), self.bins, float(lo), float(hi), _ffi.ptr(out),
                                _ffi.ptr(ws), _ffi.stream()), 'llmc_histc')
        return out.cpu().numpy()

    def add(self, sample, minmax=None):
        """One calibration sample (quant.py:471-503); minmax = its (min, max) when the caller already has them."""
        _ffi.require_gpu(sample)
        x = sample.contiguous().reshape(-1)
        if minmax is None:
            mm = torch.stack([x.min(), x.max()]).float().cpu().numpy()  # get_minmax_stats: fp32 copies of min / max
            x_min, x_max = F(mm[0]), F(mm[1])
        else:
            x_min, x_max = F(minmax[0]), F(minmax[1])
        if self.hist is None:
            self.hist, self.lo, self.hi = self._histc(x, x_min, x_max), x_min, x_max
            return
        lo, hi = min(self.lo, x_min), max(self.hi, x_max)
        upd = self._histc(x, lo, hi)
        if lo == self.lo and hi == self.hi:
            self.hist = (self.hist + upd).astype(F)
        elif self.lo == self.hi:                                          # the old histogram held a single value
            one = np.zeros(self.bins, F)
            a, b = (F(lo - 1), F(hi + 1)) if lo == hi else (lo, hi)
            one[min(int((self.lo - a) * F(self.bins) / (b - a)), self.bins - 1)] = 1
            self.hist = (one * F(np.sum(upd, dtype=F)) + upd).astype(F)
        else:
            self.hist = (upd + self._rebin(lo, hi)).astype(F)
        self.lo, self.hi = lo, hi

    # ---- histogram of the old range expressed in the bins of the new one (quant.py:333-366) ----------------------------
    @staticmethod
    def _linspace(start, end, steps):
        start, end = F(start), F(end)
        step = F((end - start) / F(steps - 1))
        i = np.arange(steps)
        return np.where(i < steps // 2, start + step * i.astype(F), end - step * (steps - 1 - i).astype(F)).astype(F)

    def _rebin(self, lo, hi):
        n = self.bins * self.ups
        fine = (np.repeat(self.hist, self.ups) / F(self.ups)).astype(F)
        half = F(0.5) * F((self.hi - self.lo) / F(n))
        mid = (self._linspace(self.lo, self.hi, n + 1)[:-1] + half).astype(F)
        edges = self._linspace(lo, hi, self.bins + 1)
        dst = np.clip(np.searchsorted(edges, mid, side='right') - 1, 0, self.bins - 1)
        out = np.zeros(self.bins, F)
        # the upsampled bins are in ascending order, so every destination bin receives a contiguous run: fp32 running
        # sums per run reproduce bincount's sequential accumulation
        start = 0
        while start < n:
            b = dst[start]
            stop = start
            acc = F(0)
            while stop < n and dst[stop] == b:
                acc = F(acc + fine[stop])
                stop += 1
            out[b] = acc
            start = stop
        return out

    # ---- error of quantizing [start_bin, end_bin] to dst uniform levels (quant.py:279-331) ------------------------------
    def _error(self, bin_width, start_bin, end_bin):
        dst_w = bin_width * (end_bin - start_bin + 1) / self.dst
        if dst_w == 0.0:
            return 0.0
        w32, d32, h = F(bin_width), F(dst_w), dst_w / 2
        begin = ((np.arange(self.bins) - start_bin).astype(F) * w32).astype(F)
        end = (begin + w32).astype(F)
        first = np.clip(np.floor(begin / d32), 0, self.dst - 1).astype(F)
        last = np.clip(np.floor(end / d32), 0, self.dst - 1).astype(F)
        density = (self.hist / w32).astype(F)

        def cube_span(b, e):          # density * (e^3 - b^3) / 3
            b, e = np.asarray(b, F), np.asarray(e, F)
            return (density * ((e * e * e - b * b * b) / F(3)).astype(F)).astype(F)

        total = cube_span((begin - ((first + F(0.5)) * d32).astype(F)).astype(F), np.full(self.bins, F(h), F))
        total = (total + ((last - first - F(1)).astype(F) * cube_span(F(-h), F(h))).astype(F)).astype(F)
        total = (total + cube_span(F(-h), (end - (last * d32 + F(h)).astype(F)).astype(F))).astype(F)
        return float(torch.from_numpy(total).sum().item())

    # ---- the search (quant.py:403-460) --------------------------------------------------------------------------------
    def range(self):
        hist = self.hist
        bin_width = (float(self.hi) - float(self.lo)) / self.bins
        total = float(torch.from_numpy(hist).sum().item())
        csum = torch.from_numpy(hist).cumsum(0).numpy()
        alpha, beta, step = 0.0, 1.0, 1e-8
        lo_bin, hi_bin, best = 0, self.bins - 1, float('inf')
        while alpha < beta:
            na, nb = alpha + step, beta - step
            left, right = lo_bin, hi_bin
            ta, tb = F(na * total), F(nb * total)
            while left < hi_bin and csum[left] < ta:
                left += 1
            while right > lo_bin and csum[right] > tb:
                right -= 1
            if (left - lo_bin) > (hi_bin - right):
                cand, alpha = (left, hi_bin), na
            else:
                cand, beta = (lo_bin, right), nb
            if cand == (lo_bin, hi_bin):
                continue
            err = self._error(bin_width, cand[0], cand[1])
            if err > best:
                break
            best, (lo_bin, hi_bin) = err, cand
        w = F((F(self.hi) - F(self.lo)) / F(self.bins))
        return F(F(self.lo) + w * F(lo_bin)), F(F(self.lo) + w * F(hi_bin + 1))


def static_hist_range(samples, bins=2048, upsample_rate=16, bit=8):
    """samples: iterable of GPU tensors (one per calibration sample) -> (min, max) as Python floats (fp32 values)."""
    h = HistRange(bins, upsample_rate, 2 ** bit)
    samples = list(samples)
    mn, mx = sample_minmax(samples)                    # every sample's range in one pass, one host copy
    mm = torch.stack([mn, mx], dim=1).cpu().numpy()
    for s, (a, b) in zip(samples, mm):
        h.add(s, minmax=(a, b))
    lo, hi = h.range()
    return float(lo), float(hi)
