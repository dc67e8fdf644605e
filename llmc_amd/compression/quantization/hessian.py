"""Hessian accumulation for GPTQ on the MFMA pipe (GPTQ.add_batch, gptq.py:254-295)."""
import ctypes as C

import torch

from llmc_amd import _ffi


class HessianAccumulator:
    """Owns H [K,K] fp32 and the partial-tile workspace; `add(inp)` has add_batch's arithmetic:
    H <- H * n/(n+b) + (2/(n+b)) * X^T X with b = number of sequences in `inp`.

    llmc's hooks call add_batch once per calibration sample (calib.bs = 1: 128 calls of [1, 2048, K] per layer). One
    SYRK launch per call would spend most of its time re-reading and re-writing H (64 MiB / 784 MiB per call) and
    filling the chip with short units, so small calls are DEFERRED: the accumulator keeps a reference to the hooked
    tensor (no copy — the activations of a block forward stay resident on the GPU anyway) and ONE launch
    walks all of them through a table of their addresses (`llmc_hessian_accum_ptrs`): the running-mean update of b
    sequences at once is the same matrix as b single updates. Reading `.H` flushes what is pending.

    A deferred tensor must not change before the flush: its version counter is checked then (pass `inp.detach()`, not
    `inp.data`, so that the counter is the producer's). `defer=False` (or the `max_pending_tokens` bound) trades the
    references for private copies / earlier launches. Inputs the kernel cannot read in place (rows not 16-B aligned, a
    strided channel axis) and very short ones (MoE experts' routed tokens) are copied compactly — their size, not a
    fixed staging buffer."""

    DIRECT_TOKENS = 16384          # calls at least this large go straight to the kernel
    SHORT_TOKENS = 256             # samples shorter than this are packed together at flush time
    MAX_PENDING_TOKENS = 1 << 19   # references one accumulator holds at most, in tokens ...
    MAX_PENDING_BYTES = 16 << 30   # ... and in bytes (tokens * K * 2): one 128 x 2048 x 28672 input (14 GiB) is still one launch
    GLOBAL_PENDING_BYTES = 48 << 30   # all accumulators of the process together (every subset of a block stays pending until its
    #                                   transform: 13 GiB for a Llama-3-8B block, 26 GiB for a 70B one); past it the adder flushes early.
    #                                   The reference frees each activation right after add_batch (gptq.py:254-295).
    _global_pending = 0            # bytes of deferred references held by all accumulators
    COPY_FLUSH_TOKENS = 65536      # defer=False: private copies are flushed at this many tokens

    def __init__(self, columns, device, defer=True, max_pending_tokens=None, exact_diag=False):
        self.K = int(columns)
        # exact_diag (GPTQ: special.hessian_exact_diag): diag(H) re-formed in fp64 by a second pass over the samples
        # (llmc_hessian_diag_accum_ptrs): the MFMA kernel's fp32 accumulation leaves 2-3e-6 of relative noise there, twice the
        # reference's sgemm, and diag(H) is what actorder sorts. One more HBM pass (2 T K bytes): off by default.
        self.exact_diag = bool(exact_diag) and self.K % 8 == 0
        self._diag64 = None
        self._diag_ws = None
        self._H = torch.zeros((self.K, self.K), dtype=torch.float32, device=device)
        self.nsamples = 0         # sequences added (pending ones included)
        self._flushed = 0         # sequences already in H
        self._ws = None
        self.defer = bool(defer)
        self.max_pending_tokens = int(max_pending_tokens or self.MAX_PENDING_TOKENS)
        self._pending = []        # (x2d, b, src, version) — src is None for private copies
        self._pending_tok = 0
        self._pending_bytes = 0
        self.timing = None   # optional list of (e0, e1, e2, T, K): e0..e1 around the MFMA kernel, e1..e2 the reduction

    @property
    def H(self):
        self.flush()
        return self._H

    @staticmethod
    def _readable_in_place(x):
        return x.stride(-1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and x.stride(0) >= x.shape[1]

    def _compact(self, x):
        T, K = x.shape
        ld = (K + 7) // 8 * 8                          # the kernel reads 16-B aligned rows
        buf = torch.empty((T, ld), dtype=x.dtype, device=x.device)[:, :K]
        buf.copy_(x)
        return buf

    def add(self, inp):
        _ffi.require_gpu(inp)
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        if inp.dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f'hessian: activations must be fp16/bf16 (model dtype), got {inp.dtype}')
        if inp.shape[-1] != self.K:
            raise ValueError(f'hessian: expected {self.K} channels, got {inp.shape[-1]}')
        viewable = True
        try:
            x = inp.view(-1, self.K)                   # a view keeps the producer's storage (and version counter)
        except RuntimeError:
            x, viewable = inp.reshape(-1, self.K), False   # not viewable as [T, K]: reshape made a private copy
        T = x.shape[0]
        if T == 0:                                     # an expert that received no token: nsamples still counts the call
            self.nsamples += b
            self._pending.append((None, b, None, 0))
            return self._H
        if T >= self.DIRECT_TOKENS:
            self.flush()
            xs = x if self._readable_in_place(x) else self._compact(x)
            self._launch([xs], b)
            self.nsamples += b
            return self._H
        if not self._readable_in_place(x) or (viewable and (not self.defer or T < self.SHORT_TOKENS)):
            self._pending.append((self._compact(x), b, None, 0))       # private compact copy
        elif not viewable:
            self._pending.append((x, b, None, 0))                      # reshape's copy is already private
        else:
            self._pending.append((x, b, inp, inp._version))            # deferred: a reference, checked at flush
        self._pending_tok += T
        nbytes = T * self.K * x.element_size()
        self._pending_bytes += nbytes
        HessianAccumulator._global_pending += nbytes
        self.nsamples += b
        if (self._pending_tok >= (self.max_pending_tokens if self.defer else self.COPY_FLUSH_TOKENS)
                or self._pending_bytes >= self.MAX_PENDING_BYTES
                or HessianAccumulator._global_pending >= self.GLOBAL_PENDING_BYTES):
            self.flush()
        return self._H

    def _drop_pending(self):
        HessianAccumulator._global_pending = max(0, HessianAccumulator._global_pending - self._pending_bytes)
        self._pending, self._pending_tok, self._pending_bytes = [], 0, 0

    def flush(self):
        """One launch per (dtype, row stride) for everything pending (no-op when nothing is)."""
        if not self._pending:
            return
        # validate BEFORE anything is dropped: a caller that catches this still holds a consistent accumulator (the pending
        # samples and nsamples agree; reset() discards them)
        for x, _, src, ver in self._pending:
            if src is not None and src._version != ver:
                raise RuntimeError('hessian: a deferred calibration tensor was modified in place before its Hessian was '
                                   'accumulated; construct the accumulator with defer=False (GPTQ: special.hessian_defer: False)')
        pend = self._pending
        self._drop_pending()
        # short samples (and whatever shares their dtype) are packed into one compact tensor per dtype
        b_total = sum(b for _, b, _, _ in pend)
        by_key, short = {}, {}
        for x, _, _, _ in pend:
            if x is None:
                continue
            if x.shape[0] < self.SHORT_TOKENS:
                short.setdefault(x.dtype, []).append(x)
            else:
                by_key.setdefault((x.dtype, x.stride(0)), []).append(x)
        for dtp, xs in short.items():
            packed = self._compact(torch.cat(xs, 0)) if len(xs) > 1 else xs[0]
            by_key.setdefault((dtp, packed.stride(0)), []).append(packed)
        groups = list(by_key.values())
        if not groups:                                  # only empty calls: H <- H * n/(n+b)
            if self._flushed:
                self._H.mul_(self._flushed / (self._flushed + b_total))
                if self._diag64 is not None:
                    self._diag64.mul_(self._flushed / (self._flushed + b_total))
            else:
                self._H.zero_()                         # after reset() the buffer still holds the previous Hessian: 0 * n/(n+b)
                if self._diag64 is not None:
                    self._diag64.zero_()
            self._flushed += b_total
            return
        # the sequences of the whole flush enter the running mean with the first launch; the others add their products
        first = True
        for xs in groups:
            self._launch(xs, b_total if first else 0)
            first = False

    def _launch(self, xs, b):
        """xs: [T_i, K] tensors of one dtype and one row stride; b sequences enter the running mean with this launch."""
        L = _ffi.lib()
        nmax = L.llmc_hessian_max_samples()
        for i in range(0, len(xs), nmax):
            self._launch_some(xs[i:i + nmax], b if i == 0 else 0)

    def _launch_some(self, xs, b):
        L = _ffi.lib()
        n, K, ldx = len(xs), self.K, xs[0].stride(0)
        Ts = (C.c_int64 * n)(*[x.shape[0] for x in xs])
        Xs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        need = L.llmc_hessian_accum_ptrs_ws_bytes(Ts, n, K, ldx)
        if need == 0:
            # the samples are too short / too many for one walk: pack them into one tensor
            xs = [self._compact(torch.cat(xs, 0))]
            n, ldx = 1, xs[0].stride(0)
            Ts = (C.c_int64 * 1)(xs[0].shape[0])
            Xs = (C.c_void_p * 1)(xs[0].data_ptr())
            need = L.llmc_hessian_accum_ptrs_ws_bytes(Ts, n, K, ldx)
            if need == 0:
                _ffi.check(-22, 'llmc_hessian_accum_ptrs_ws_bytes')
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = _ffi.workspace(need, xs[0].device)
        st = _ffi.stream()
        T = sum(x.shape[0] for x in xs)
        if self.timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _ffi.check(L.llmc_hessian_accum_ptrs_partials(Xs, Ts, n, _ffi.dt(xs[0]), K, ldx, _ffi.ptr(self._ws), st),
                   'llmc_hessian_accum_ptrs_partials')
        if self.timing is not None:
            e1.record()
        # b = 0: a further launch of the same flush adds its products with the weights of the first (n stays)
        nb, na = float(self._flushed), float(self._flushed + b)
        if b == 0:
            nb = na = float(self._flushed)
        _ffi.check(L.llmc_hessian_accum_ptrs_reduce(_ffi.ptr(self._H), Ts, n, K, ldx, nb, na, _ffi.ptr(self._ws), st),
                   'llmc_hessian_accum_ptrs_reduce')
        if self.timing is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            self.timing.append((e0, e1, e2, T, K))
        if self.exact_diag:
            if self._diag64 is None:
                self._diag64 = torch.zeros(K, dtype=torch.float64, device=xs[0].device)
                self._diag_ws = _ffi.workspace(L.llmc_hessian_diag_ws_bytes(K), xs[0].device)
            _ffi.check(L.llmc_hessian_diag_accum_ptrs(_ffi.ptr(self._H), _ffi.ptr(self._diag64), Xs, Ts, n, _ffi.dt(xs[0]), K, ldx,
                                                      nb, na, _ffi.ptr(self._diag_ws), st), 'llmc_hessian_diag_accum_ptrs')
        self._flushed += b
        self._last_launch = (Ts, n, K, ldx)
        # the tensors of xs may be released by the caller once this returns: the launches are stream-ordered and torch's
        # allocator keeps a freed block out of other streams' hands until this stream has passed

    def barrier_timeouts(self):
        """Diagnostic (synchronises the current stream): round barriers of the most recent SYRK launch that gave up waiting
        for workgroups other streams kept off their CUs (llmc_hessian_accum_barrier_timeouts). 0 in a healthy run."""
        last = getattr(self, '_last_launch', None)
        if last is None or self._ws is None:
            return 0
        Ts, n, K, ldx = last
        out = C.c_uint(0)
        _ffi.check(_ffi.lib().llmc_hessian_accum_barrier_timeouts(_ffi.ptr(self._ws), Ts, n, K, ldx, C.byref(out), _ffi.stream()),
                   'llmc_hessian_accum_barrier_timeouts')
        return int(out.value)

    def reset(self):
        """Start a new Hessian in the same buffers (the first launch overwrites H: n_before = 0)."""
        self.nsamples = 0
        self._flushed = 0
        if self._diag64 is not None:
            self._diag64.zero_()        # the fp64 running diagonal belongs to the Hessian that is being discarded
        self._drop_pending()

    def release_workspace(self):
        self._ws = None
        self._drop_pending()

    def __del__(self):
        try:
            self._drop_pending()
        except Exception:       # noqa: BLE001 (interpreter shutdown)
            pass
