"""Hessian accumulation for GPTQ on the MFMA pipe (GPTQ.add_batch, gptq.py:254-295)."""
import torch

from llmc_amd import _ffi


class HessianAccumulator:
    """Owns H [K,K] fp32 and the partial-tile workspace; `add(inp)` has add_batch's arithmetic:
    H <- H * n/(n+b) + (2/(n+b)) * X^T X with b = number of sequences in `inp`.

    llmc's hooks call add_batch once per calibration sample (calib.bs = 1: 128 calls of 2048 tokens per layer). One
    SYRK launch per call would spend most of its time re-reading and re-writing H (64 MiB / 784 MiB per call) and
    filling the chip with 32-K-step units, so small calls are STAGED: their tokens are copied into a resident
    [stage_tokens, K] buffer and one launch covers up to `stage_tokens` of them — the running-mean update of b
    sequences at once is the same matrix as b single updates. Reading `.H` flushes what is staged."""

    STAGE_TOKENS = 65536          # 32 sequences of 2048 tokens: 512 MiB at K = 4096, 1.75 GiB at K = 14336 (16-bit)
    DIRECT_TOKENS = 16384         # calls at least this large go straight to the kernel

    def __init__(self, columns, device):
        self.K = int(columns)
        self._H = torch.zeros((self.K, self.K), dtype=torch.float32, device=device)
        self.nsamples = 0         # sequences added (staged ones included)
        self._flushed = 0         # sequences already in H
        self._ws = None
        self._stage = None
        self._stage_tok = 0
        self.timing = None   # optional list of (e0, e1, e2, T, K): e0..e1 around the MFMA kernel, e1..e2 the reduction

    @property
    def H(self):
        self.flush()
        return self._H

    def add(self, inp):
        _ffi.require_gpu(inp)
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        x = inp.reshape(-1, inp.shape[-1])
        if x.dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f'hessian: activations must be fp16/bf16 (model dtype), got {x.dtype}')
        T, K = x.shape
        if K != self.K:
            raise ValueError(f'hessian: expected {self.K} channels, got {K}')
        if T >= self.DIRECT_TOKENS or T > self.STAGE_TOKENS:
            self.flush()
            self._launch(x, b)
            self.nsamples += b
            return self._H
        if self._stage is not None and (self._stage.dtype != x.dtype or self._stage_tok + T > self.STAGE_TOKENS):
            self.flush()
        if self._stage is None or self._stage.dtype != x.dtype:
            ld = (K + 7) // 8 * 8                      # the kernel reads 16-B aligned rows
            self._stage = torch.empty((self.STAGE_TOKENS, ld), dtype=x.dtype, device=x.device)[:, :K]
        self._stage[self._stage_tok:self._stage_tok + T].copy_(x)
        self._stage_tok += T
        self.nsamples += b
        return self._H

    def flush(self):
        """One launch for everything staged (no-op when nothing is)."""
        if self._stage_tok:
            b = self.nsamples - self._flushed
            self._launch(self._stage[:self._stage_tok], b, staged=True)
            self._stage_tok = 0

    def _launch(self, x, b, staged=False):
        L = _ffi.lib()
        T, K = x.shape
        if x.stride(-1) != 1 or x.stride(0) % 8 != 0 or x.data_ptr() % 16 != 0:
            ld = (K + 7) // 8 * 8
            buf = torch.empty((T, ld), dtype=x.dtype, device=x.device)[:, :K]
            buf.copy_(x)
            x = buf
        ldx = x.stride(0)
        need = L.llmc_hessian_accum_ws_bytes(T, K, ldx)
        if self._ws is None or self._ws.numel() < need:
            self._ws = _ffi.workspace(need, x.device)
        st = _ffi.stream()
        if self.timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _ffi.check(L.llmc_hessian_accum_partials(_ffi.ptr(x), _ffi.dt(x), T, K, ldx, _ffi.ptr(self._ws), st),
                   'llmc_hessian_accum_partials')
        if self.timing is not None:
            e1.record()
        _ffi.check(L.llmc_hessian_accum_reduce(_ffi.ptr(self._H), T, K, ldx, float(self._flushed),
                                               float(self._flushed + b), _ffi.ptr(self._ws), st),
                   'llmc_hessian_accum_reduce')
        if self.timing is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            self.timing.append((e0, e1, e2, T, K))
        self._flushed += b

    def reset(self):
        """Start a new Hessian in the same buffers (the first launch overwrites H: n_before = 0)."""
        self.nsamples = 0
        self._flushed = 0
        self._stage_tok = 0

    def release_workspace(self):
        self._ws = None
        self._stage = None
