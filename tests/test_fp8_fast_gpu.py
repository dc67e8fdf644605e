"""The division-free form of the FP8 cast (csrc/fp8_pack.hip:fp8_fast8) against the same kernel with FP8_EXACT_DIV (every element
through the IEEE division and the general encoder): identical codes and identical fake-quantized values, for both semantics,
bf16 and fp16, per_tensor and per_channel, on tensors large enough to hit every guard thousands of times."""
import time

import numpy as np
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu
EXACT_DIV = 0x200


def run(w, G, g, mode, sdt):
    from llmc_amd import _ffi
    L = _ffi.lib()
    fake = mode & 1
    out = torch.empty_like(w) if fake else torch.empty(w.shape, dtype=torch.uint8, device=w.device)
    s = torch.empty(G, dtype=sdt, device=w.device)
    ws = _ffi.workspace(L.llmc_fp8_quant_ws_bytes(G, g), w.device)
    _ffi.check(L.llmc_fp8_quant(_ffi.ptr(w), _ffi.dt(w), G, g, mode, _ffi.ptr(out), _ffi.ptr(s), _ffi.dt(sdt), 0, _ffi.ptr(ws),
                                _ffi.stream()), 'llmc_fp8_quant')
    return out, s


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('sem', [0x100, 0])
def test_fast_cast_equals_the_exact_division_path(dt, sem):
    gen = torch.Generator().manual_seed(5 + sem)
    R, K = 4096, 4096
    w = torch.randn(R, K, generator=gen) * 0.03
    w[torch.rand(R, K, generator=gen) < 1e-3] *= 6
    w[:, torch.randperm(K, generator=gen)[:4]] *= 20      # outlier channels: the per-tensor scale pushes ~0.3 % of the elements below 2^-6
    w[0, :64] = 0.0
    w[1, :64] = -0.0
    w[2, :64] = 1e-7                      # the 8-bit format's subnormal range after scaling
    w[3, :64] = 6e-8 if dt == torch.float16 else 1e-30
    w = w.to(dt).cuda()
    for gran, G, g, sdt in (('per_tensor', 1, R * K, torch.float32), ('per_channel', R, K, dt)):
        for fake in (0, 1):
            mode = fake | sem
            a, sa = run(w, G, g, mode, sdt)
            b, sb = run(w, G, g, mode | EXACT_DIV, sdt)
            assert torch.equal(sa, sb)
            if fake:
                assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (gran, fake, float((a != b).float().mean()))
            else:
                assert torch.equal(a, b), (gran, fake, float((a != b).float().mean()))


def test_cast_rate_at_the_mixtral_expert_shape():
    """14336 x 4096 bf16 per_tensor (BASELINE configs[4]): absmax + cast, codes; reported, with a loose floor."""
    from llmc_amd.compression.quantization import FloatQuantizer
    w = (torch.randn(14336, 4096, device='cuda') * 0.02).to(torch.bfloat16)
    for sem in ('qtorch', 'cast'):
        q = FloatQuantizer('e4m3', True, 'per_tensor', use_qtorch=True, fp8_semantics=sem)
        for _ in range(3):
            q.real_quant_weight_dynamic(w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            q.real_quant_weight_dynamic(w)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        report('fp8_per_tensor_14336x4096/' + sem, us=us, tb_per_s_at_3B_per_element=3 * w.numel() / us / 1e6)
        assert us < 150
