"""K1 hessian_syrk (MFMA) vs the oracle / fp64 ground truth."""
import numpy as np
import pytest
import torch

from oracle import gptq_ref as G

pytestmark = pytest.mark.gpu
TD = {'f16': torch.float16, 'bf16': torch.bfloat16}


def make_x(T, K, dt, seed, b=1):
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(b, T, K, generator=gen)
    c = torch.exp(0.5 * torch.randn(K, generator=gen))
    x = z * c
    x[..., :: max(1, K // 7)] *= 30.0   # outlier channels
    return x.to(TD[dt])


def rel_err(h, ref):
    d = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-30
    return np.abs(h - ref) / d


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('shape', [(200, 304), (64, 256), (1000, 768), (130, 1032)])
def test_small_shapes_vs_fp64(dt, shape):
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    T, K = shape
    batches = [make_x(T, K, dt, 10 + i) for i in range(3)]
    acc = HessianAccumulator(K, 'cuda')
    Href = np.zeros((K, K), dtype=np.float32)
    n = 0
    for b in batches:
        acc.add(b.cuda())
        Href, n = G.add_batch(Href, n, b.float().numpy())
    H = acc.H.cpu().numpy()
    exact = G.hessian_exact([b.float().numpy() for b in batches])
    assert acc.nsamples == 3
    np.testing.assert_array_equal(H, H.T)                       # symmetric by construction
    e_ours = rel_err(H, exact).max()
    e_ref = rel_err(Href, exact).max()
    # fp32 accumulation: we must be at least as close to the truth as the reference's own fp32 path (x4 slack)
    assert e_ours <= max(4 * e_ref, 2e-6), (e_ours, e_ref)
    assert rel_err(H, Href).max() < 2e-5


def test_strided_rows_and_asymmetric_pattern():
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    T, K, ld = 96, 300, 320
    base = torch.zeros(T, ld, dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(3)
    base[:, :K] = (torch.randint(-3, 4, (T, K), generator=gen)).to(torch.bfloat16)
    base[:, K:] = 1000.0   # padding must never leak into H
    xd = base.cuda()[:, :K]
    acc = HessianAccumulator(K, 'cuda')
    acc.add(xd)
    x = base[:, :K].double().numpy()
    exact = 2.0 * x.T @ x
    np.testing.assert_array_equal(acc.H.cpu().numpy(), exact.astype(np.float32))  # small ints: exact


@pytest.mark.parametrize('dt', ['bf16'])
def test_llama_k4096_chunked_and_deterministic(dt):
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    T, K = 8192, 4096
    x = make_x(T, K, dt, 99, b=4).reshape(4, T, K)
    xd = x.cuda()
    acc = HessianAccumulator(K, 'cuda')
    acc.add(xd)                       # one call, 4 sequences (calib.bs = 4)
    H1 = acc.H.clone()
    acc2 = HessianAccumulator(K, 'cuda')
    acc2.add(xd)
    assert torch.equal(H1, acc2.H)    # launch-independent summation order
    # fp64 ground truth at full width; torch's own fp32 GEMM on the same data sets the error scale
    xf = xd.reshape(-1, K)
    ref = ((xf.double().T @ xf.double()) * (2.0 / 4))
    d = torch.sqrt(torch.outer(torch.diag(ref), torch.diag(ref)))
    ref32 = (xf.float().T @ xf.float()) * (2.0 / 4)
    e_t = ((ref32.double() - ref).abs() / d).max().item()
    e_1 = ((H1.double() - ref).abs() / d).max().item()
    assert e_1 <= max(4 * e_t, 2e-6), (e_1, e_t)
    # running mean over 4 separate hook calls (calib.bs = 1) converges to the same matrix
    acc3 = HessianAccumulator(K, 'cuda')
    for i in range(4):
        acc3.add(xd[i])
    e_3 = ((acc3.H.double() - ref).abs() / d).max().item()
    assert e_3 <= max(4 * e_t, 2e-6), (e_3, e_t)


def test_cu_reserve_leaves_compute_units_free_and_changes_no_bit():
    """llmc_hip_set_cu_reserve: the persistent Hessian kernel launches on fewer compute units (the caller keeps the rest
    for kernels of other streams); chunking does not depend on the grid, so H is bit-identical. The setter is per
    thread and returns the previous value."""
    from llmc_amd import _ffi
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    x = make_x(3000, 1024, 'bf16', 3).cuda()
    a = HessianAccumulator(1024, 'cuda')
    a.add(x)
    h0 = a.H.clone()
    assert _ffi.lib().llmc_hip_set_cu_reserve(0) == 0
    with _ffi.cu_reserve(64):
        b = HessianAccumulator(1024, 'cuda')
        b.add(x)
        h1 = b.H.clone()
        assert _ffi.lib().llmc_hip_set_cu_reserve(64) == 64
    assert _ffi.lib().llmc_hip_set_cu_reserve(0) == 0
    assert torch.equal(h0, h1)
