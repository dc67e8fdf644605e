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

    def __init__(self, columns, device, defer=True, max_pending_tokens=None, exact_diag=True):
        self.K = int(columns)
        # exact_diag (GPTQ: special.hessian_exact_diag, default True since round 6): diag(H) — what actorder sorts and the damping
        # averages — is folded into fp64 every 128 tokens by the MFMA kernel's otherwise redundant diagonal-tile wave
        # (hessian_syrk.hip: no extra pass over the samples) and keeps its unrounded running value in `_diag64` across launches.
        # False: the fp32 chain's own diagonal (2-3e-6 of relative noise, twice the reference's sgemm), for A/B.
        self.exact_diag = bool(exact_diag)
        self._diag64 = None
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

    def _take_pending(self):
        """Validate and take everything pending: returns (launch groups [[x, ...], ...] — one per (dtype, row stride) —, b_total).
        Nothing is dropped when the validation raises."""
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
        return list(by_key.values()), b_total

    def flush(self):
        """One launch per (dtype, row stride) for everything pending (no-op when nothing is)."""
        if not self._pending:
            return
        groups, b_total = self._take_pending()
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

    @staticmethod
    def flush_many(accs, mix_widths=False):
        """Flush several accumulators together: those whose pending samples form ONE launch each (one dtype, one row stride —
        the hook calls of a block forward) share launches of up to llmc_hessian_max_problems() Hessians of the same width
        (llmc_hessian_accum_multi_*: one unit queue, the triangular tails fill rounds together — the three K = 4096 inputs
        of a Llama block); everything else is flushed on its own. The token-chunk count of a launch is chosen for the problems that
        share it, so against one-by-one launches the fp32 sums are formed in another order (summation-order noise, <= 1e-6).
        mix_widths: Hessians of different widths share a launch too (measured: profiles/r06_k1_ab.txt)."""
        L = _ffi.lib()
        pmax, nmax = L.llmc_hessian_max_problems(), L.llmc_hessian_max_samples()
        singles, seen = {}, set()
        for a in accs:
            if id(a) in seen or not a._pending:
                continue
            seen.add(id(a))
            groups, b_total = a._take_pending()
            if len(groups) == 1 and len(groups[0]) <= nmax // 2:
                xs = groups[0]
                singles.setdefault((None if mix_widths else a.K, xs[0].dtype, xs[0].device), []).append((a, xs, b_total))
                continue
            # not one launch: put it back the simple way
            if not groups:
                a._pending = [(None, b_total, None, 0)]
                a.flush()
                continue
            first = True
            for xs in groups:
                a._launch(xs, b_total if first else 0)
                first = False
        for items in singles.values():
            items.sort(key=lambda it: -it[0].K)      # mix_widths: the widest first, so that the last round holds the shortest units
            batch, nsmp = [], 0
            for it in items + [None]:
                if it is None or len(batch) == pmax or nsmp + len(it[1]) > nmax:
                    if len(batch) == 1:
                        batch[0][0]._launch(batch[0][1], batch[0][2])
                    elif batch:
                        HessianAccumulator._launch_multi(batch)
                    batch, nsmp = [], 0
                if it is not None:
                    batch.append(it)
                    nsmp += len(it[1])

    def _launch(self, xs, b):
        """xs: [T_i, K] tensors of one dtype and one row stride; b sequences enter the running mean with this launch."""
        L = _ffi.lib()
        nmax = L.llmc_hessian_max_samples()
        for i in range(0, len(xs), nmax):
            self._launch_some(xs[i:i + nmax], b if i == 0 else 0)

    def _problem(self, pr, xs, b):
        """Fill `pr` (llmc_hessian_problem_t) for one launch of this accumulator; returns the ctypes arrays it points at (the
        caller keeps them alive until the call has returned)."""
        n, K, ldx = len(xs), self.K, xs[0].stride(0)
        Ts = (C.c_int64 * n)(*[x.shape[0] for x in xs])
        Xs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        # b = 0: a further launch of the same flush adds its products with the weights of the first (n stays)
        nb, na = float(self._flushed), float(self._flushed + b)
        if b == 0:
            nb = na = float(self._flushed)
        if self.exact_diag and self._diag64 is None:
            self._diag64 = torch.zeros(K, dtype=torch.float64, device=xs[0].device)
        pr.H, pr.dstate = _ffi.ptr(self._H), (_ffi.ptr(self._diag64) if self.exact_diag else None)
        pr.X_list_host, pr.T_list_host = C.cast(Xs, C.c_void_p), C.cast(Ts, C.c_void_p)
        pr.n, pr.K, pr.ldx, pr.n_before, pr.n_after = n, K, ldx, nb, na
        return Ts, Xs

    def _launch_some(self, xs, b):
        L = _ffi.lib()
        arr = (_ffi.HessianProblem * 1)()
        keep = self._problem(arr[0], xs, b)
        need = L.llmc_hessian_accum_multi_ws_bytes(arr, 1)
        if need == 0:
            # the samples are too short / too many for one walk: pack them into one tensor
            xs = [self._compact(torch.cat(xs, 0))]
            keep = self._problem(arr[0], xs, b)
            need = L.llmc_hessian_accum_multi_ws_bytes(arr, 1)
            if need == 0:
                _ffi.check(-22, 'llmc_hessian_accum_multi_ws_bytes')
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = None
            self._ws = _ffi.workspace(need + 256, xs[0].device)
        HessianAccumulator._run([self], arr, 1, [xs], [b], self._ws, self.timing, keep)

    @staticmethod
    def _launch_multi(batch):
        """batch: [(accumulator, xs, b)] of one width / dtype: ONE launch pair for all of them."""
        L = _ffi.lib()
        P = len(batch)
        arr = (_ffi.HessianProblem * P)()
        keep = [a._problem(arr[i], xs, b) for i, (a, xs, b) in enumerate(batch)]
        need = L.llmc_hessian_accum_multi_ws_bytes(arr, P)
        if need == 0:
            for a, xs, b in batch:
                a._launch(xs, b)
            return
        owner = batch[0][0]
        if owner._ws is None or owner._ws.numel() < need + 256:
            owner._ws = None
            owner._ws = _ffi.workspace(need + 256, batch[0][1][0].device)
        HessianAccumulator._run([a for a, _, _ in batch], arr, P, [xs for _, xs, _ in batch], [b for _, _, b in batch], owner._ws,
                                owner.timing, keep)

    @staticmethod
    def _run(accs, probs, P, xs_list, bs, ws, timing, keep):
        L = _ffi.lib()
        st = _ffi.stream()
        dtc = _ffi.dt(xs_list[0][0])
        wsp = ws.data_ptr() + ((-ws.data_ptr()) % 256)
        fp32_diag = [a for a in accs if not a.exact_diag]
        if fp32_diag and len(fp32_diag) != len(accs):
            raise ValueError('hessian: accumulators with and without exact_diag cannot share a launch')
        with _ffi.option(k1_fp32_diag=1 if fp32_diag else 0):       # both calls of the pair see the same switch
            if timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _ffi.check(L.llmc_hessian_accum_multi_partials(probs, P, dtc, wsp, st), 'llmc_hessian_accum_multi_partials')
            if timing is not None:
                e1.record()
            _ffi.check(L.llmc_hessian_accum_multi_reduce(probs, P, wsp, st), 'llmc_hessian_accum_multi_reduce')
        if timing is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            flops = sum(sum(x.shape[0] for x in xs) * a.K * (a.K + 1) for a, xs in zip(accs, xs_list))
            timing.append((e0, e1, e2, flops, P))
        for a, b in zip(accs, bs):
            a._flushed += b
        for a in accs:
            a._last_launch = (probs, P, ws, keep)
        # the tensors of xs may be released by the caller once this returns: the launches are stream-ordered and torch's
        # allocator keeps a freed block out of other streams' hands until this stream has passed

    def barrier_timeouts(self):
        """Diagnostic (synchronises the current stream): round barriers of the most recent SYRK launch that gave up waiting
        for workgroups other streams kept off their CUs (llmc_hessian_accum_multi_barrier_timeouts). 0 in a healthy run."""
        last = getattr(self, '_last_launch', None)
        if last is None:
            return 0
        probs, P, ws, _keep = last
        out = C.c_uint(0)
        wsp = ws.data_ptr() + ((-ws.data_ptr()) % 256)
        _ffi.check(_ffi.lib().llmc_hessian_accum_multi_barrier_timeouts(probs, P, wsp, C.byref(out), _ffi.stream()),
                   'llmc_hessian_accum_multi_barrier_timeouts')
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
