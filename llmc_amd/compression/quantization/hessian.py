"""Hessian accumulation for GPTQ on the MFMA pipe (GPTQ.add_batch, gptq.py:254-295)."""
import torch

from llmc_amd import _ffi


class HessianAccumulator:
    """Owns H [K,K] fp32 and the partial-tile workspace; `add(inp)` has add_batch's arithmetic:
    H <- H * n/(n+b) + (2/(n+b)) * X^T X with b = number of sequences in `inp`."""

    def __init__(self, columns, device):
        self.K = int(columns)
        self.H = torch.zeros((self.K, self.K), dtype=torch.float32, device=device)
        self.nsamples = 0
        self._ws = None
        self.timing = None   # optional list of (e0, e1, e2, T, K): e0..e1 around the MFMA kernel, e1..e2 the reduction

    def add(self, inp):
        _ffi.require_gpu(inp)
        L = _ffi.lib()
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        x = inp.reshape(-1, inp.shape[-1])
        if x.dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f'hessian: activations must be fp16/bf16 (model dtype), got {x.dtype}')
        if x.stride(-1) != 1 or x.stride(0) % 8 != 0 or x.data_ptr() % 16 != 0:
            x = x.contiguous()
        T, K = x.shape
        if K != self.K:
            raise ValueError(f'hessian: expected {self.K} channels, got {K}')
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
        _ffi.check(L.llmc_hessian_accum_reduce(_ffi.ptr(self.H), T, K, ldx, float(self.nsamples),
                                               float(self.nsamples + b), _ffi.ptr(self._ws), st),
                   'llmc_hessian_accum_reduce')
        if self.timing is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            self.timing.append((e0, e1, e2, T, K))
        self.nsamples += b
        return self.H

    def reset(self):
        """Start a new Hessian in the same buffers (the first add() overwrites H: n_before = 0)."""
        self.nsamples = 0

    def release_workspace(self):
        self._ws = None
