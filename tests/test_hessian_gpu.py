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


def _fp64_hessian(samples, n_batches):
    K = samples[0].shape[-1]
    acc = torch.zeros(K, K, dtype=torch.float64, device='cuda')
    for smp in samples:
        xf = smp.reshape(-1, K).double()
        acc += xf.T @ xf
    return acc * (2.0 / n_batches)


def _check_vs_fp64(H, samples, n_batches):
    ref = _fp64_hessian(samples, n_batches)
    d = torch.sqrt(torch.outer(torch.diag(ref), torch.diag(ref))) + 1e-30
    K = ref.shape[0]
    xf = torch.cat([s.reshape(-1, K) for s in samples], 0).float()
    e_t = (((xf.T @ xf) * (2.0 / n_batches)).double() - ref).abs().div(d).max().item()
    e = ((H.double() - ref).abs() / d).max().item()
    assert e <= max(4 * e_t, 2e-6), (e, e_t)
    assert torch.equal(H, H.T)


def test_sample_table_equals_one_tensor_bitwise():
    """llmc_hessian_accum_ptrs over separately allocated samples (lengths multiples of 128) == llmc_hessian_accum over
    their concatenation, bit for bit; defer=False (private copies, same table walk) as well."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    K, seq, n = 1280, 512, 24
    x = make_x(seq, K, 'bf16', 5, b=n).cuda()
    one = HessianAccumulator(K, 'cuda')
    one.add(x)                                              # 12288 tokens < DIRECT_TOKENS: a one-entry table
    H1 = one.H.clone()
    samples = [x[i:i + 1].clone() for i in range(n)]
    for defer in (True, False):
        acc = HessianAccumulator(K, 'cuda', defer=defer)
        acc.timing = []
        for smp in samples:
            acc.add(smp)
        assert torch.equal(acc.H, H1), defer
        assert len(acc.timing) == 1 and acc.nsamples == n
    # the C entry point with the flat signature agrees as well
    from llmc_amd import _ffi
    L = _ffi.lib()
    xf = x.reshape(-1, K)
    ws = _ffi.workspace(L.llmc_hessian_accum_ws_bytes(xf.shape[0], K, K), xf.device)
    H0 = torch.empty(K, K, dtype=torch.float32, device='cuda')
    _ffi.check(L.llmc_hessian_accum(_ffi.ptr(H0), _ffi.ptr(xf), _ffi.dt(xf), xf.shape[0], K, K, 0.0, float(n), _ffi.ptr(ws),
                                    _ffi.stream()), 'llmc_hessian_accum')
    assert torch.equal(H0, H1)


def test_sample_table_ragged_lengths_and_many_samples():
    """Samples of any length (each walked in 128-token groups, the last one zero-filled by its own descriptor), more
    samples than one launch takes (several launches, one running mean), short samples (packed), an empty call."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    K = 776
    gen = torch.Generator().manual_seed(11)
    lens = [300, 1000, 257, 4096, 129, 640, 2047, 511] + [260 + (3 * i) % 200 for i in range(540)] + [5, 17, 200, 64]
    samples = [(torch.randn(1, t, K, generator=gen) * 3).to(torch.bfloat16).cuda() for t in lens]
    acc = HessianAccumulator(K, 'cuda')
    acc.timing = []
    for smp in samples:
        acc.add(smp)
    acc.add(torch.empty(1, 0, K, dtype=torch.bfloat16, device='cuda'))      # counts as a batch, adds nothing
    H = acc.H
    assert acc.nsamples == len(lens) + 1 and len(acc.timing) >= 2
    _check_vs_fp64(H, samples, len(lens) + 1)
    # running mean across flushes: two flushes == what the reference's per-call update converges to
    acc2 = HessianAccumulator(K, 'cuda')
    for smp in samples[:100]:
        acc2.add(smp)
    acc2.flush()
    for smp in samples[100:]:
        acc2.add(smp)
    _check_vs_fp64(acc2.H, samples, len(lens))


def test_sample_table_strided_rows_in_place_and_padding_never_read():
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    T, K, ld = 640, 300, 320
    gen = torch.Generator().manual_seed(3)
    bases = []
    for i in range(3):
        base = torch.zeros(T, ld, dtype=torch.bfloat16)
        base[:, :K] = (torch.randint(-3, 4, (T, K), generator=gen)).to(torch.bfloat16)
        base[:, K:] = 1000.0   # padding must never leak into H
        bases.append(base.cuda())
    acc = HessianAccumulator(K, 'cuda')
    for b in bases:
        acc.add(b[:, :K])                                   # read in place: row stride 320, 300 channels
    assert all(src is not None for _, _, src, _ in acc._pending)
    x = torch.cat([b[:, :K] for b in bases], 0).double().cpu().numpy()
    exact = (2.0 / 3) * x.T @ x
    np.testing.assert_allclose(acc.H.cpu().numpy(), exact, rtol=2e-6, atol=1e-4)


def test_deferred_sample_modified_before_flush_raises():
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    K = 512
    x = make_x(512, K, 'bf16', 1).cuda()
    acc = HessianAccumulator(K, 'cuda')
    acc.add(x)
    x.mul_(2)                                               # the producer reuses the buffer
    with pytest.raises(RuntimeError):
        acc.H


def test_failed_flush_keeps_the_pending_samples_and_reset_then_empty_calls_start_from_zero():
    """(a) a flush that raises (a deferred tensor was modified) must not drop the pending samples while nsamples
    still counts them; (b) after reset(), a first flush that holds only EMPTY calls (an expert without tokens) must not let the
    previous Hessian leak into the new one through alpha = n/(n+b)."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    K = 256
    acc = HessianAccumulator(K, 'cuda')
    xs = [make_x(300, K, 'bf16', 40 + i).cuda() for i in range(3)]
    for x in xs:
        acc.add(x)
    xs[1].mul_(2.0)
    with pytest.raises(RuntimeError):
        acc.flush()
    assert acc.nsamples == 3 and len(acc._pending) == 3 and acc._pending_bytes == 3 * 300 * K * 2
    # (b): fill H, reset, then an empty call, a flush, and one real sample: H must be that sample's Hessian alone
    acc.reset()
    assert len(acc._pending) == 0 and acc._pending_bytes == 0
    acc.add(xs[0])
    assert float(acc.H.abs().max()) > 0
    acc.reset()
    acc.add(torch.empty((1, 0, K), dtype=torch.bfloat16, device='cuda'))
    acc.flush()
    assert float(acc._H.abs().max()) == 0.0
    acc.add(xs[2])
    H = acc.H.cpu().numpy()
    Href, n = G.add_batch(np.zeros((K, K), np.float32), 1, xs[2].float().cpu().numpy())   # one empty sample came first: n = 1
    assert acc.nsamples == 2 and n == 2
    assert rel_err(H, Href).max() < 1e-5


def test_pending_references_are_bounded_in_bytes(monkeypatch):
    """the bound on deferred references counts bytes (tokens x K x 2), per accumulator and across accumulators."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    K = 512
    monkeypatch.setattr(HessianAccumulator, 'MAX_PENDING_BYTES', 3 * 400 * K * 2)
    a = HessianAccumulator(K, 'cuda')
    xs = [make_x(400, K, 'bf16', 50 + i).cuda() for i in range(4)]
    a.add(xs[0]); a.add(xs[1])
    assert len(a._pending) == 2
    a.add(xs[2])                                    # reaches the per-accumulator bound: flushed
    assert len(a._pending) == 0 and a._flushed == 3
    monkeypatch.setattr(HessianAccumulator, 'MAX_PENDING_BYTES', 1 << 40)
    monkeypatch.setattr(HessianAccumulator, 'GLOBAL_PENDING_BYTES', HessianAccumulator._global_pending + 3 * 400 * K * 2)
    b = HessianAccumulator(K, 'cuda')
    a.add(xs[3]); b.add(xs[0])
    assert len(a._pending) == 1 and len(b._pending) == 1
    b.add(xs[1])                                    # the global budget: the adder flushes
    assert len(b._pending) == 0 and len(a._pending) == 1
    ref = HessianAccumulator(K, 'cuda', defer=False)
    for x in xs:
        ref.add(x)
    # an early flush splits the running mean into two updates: the same matrix up to fp32 rounding
    assert rel_err(a.H.cpu().numpy(), ref.H.cpu().numpy()).max() < 1e-5


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_exact_diag_option_gives_the_fp64_diagonal_and_leaves_the_rest_alone(dt):
    """The default (round 6): diag(H) = (2 / n) sum x^2 folded into fp64 every 256 tokens by the MFMA kernel's diagonal-tile wave
    and rounded once — at the fp32 rounding level of the exact value, where an fp32 chain over thousands of tokens
    (exact_diag=False, rounds 1-5's default) is 10x further away; every off-diagonal bit is the same either way; per-sample
    feeds, several flushes (running mean) and ragged sample lengths included."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    K = 1024
    gen = torch.Generator(device='cuda').manual_seed(11)
    xs = [(torch.randn(1, t, K, generator=gen, device='cuda') * torch.exp(0.7 * torch.randn(K, generator=gen, device='cuda'))).to(dt)
          for t in (2048, 2048, 777, 2048, 300, 2048, 4096, 2048)]
    a0 = HessianAccumulator(K, 'cuda', exact_diag=False)
    a1 = HessianAccumulator(K, 'cuda')
    for i, x in enumerate(xs):
        a0.add(x)
        a1.add(x)
        if i == 3:                      # a flush in the middle: the running mean continues (n_before > 0)
            _ = a0.H, a1.H
    H0, H1 = a0.H.clone(), a1.H.clone()
    off = ~torch.eye(K, dtype=torch.bool, device='cuda')
    assert torch.equal(H0[off], H1[off])
    ref = sum((x.double() ** 2).sum(dim=(0, 1)) for x in xs) * (2.0 / len(xs))
    e1 = ((torch.diagonal(H1).double() - ref).abs() / ref).max().item()
    e0 = ((torch.diagonal(H0).double() - ref).abs() / ref).max().item()
    assert e1 <= 1.2e-7, e1             # half an fp32 ulp (6e-8) + the fp32 chains of 16 MFMAs between the fp64 folds
    assert e0 > e1                      # (informative: the default diagonal is the noisier one)
    # reset() starts over
    a1.reset()
    a1.add(xs[0])
    r0 = (xs[0].double() ** 2).sum(dim=(0, 1)) * 2.0
    # one 2048-token sample = eight fp64 folds of 16 chained MFMAs each: their fp32 rounding (1e-7) is not averaged down yet
    assert ((torch.diagonal(a1.H).double() - r0).abs() / r0).max().item() <= 2.5e-7


@pytest.mark.parametrize('K,lens', [(4096, [2048] * 6), (1024, [777, 2048, 300, 1536]), (640, [2048, 2048])])
def test_several_hessians_in_one_launch(K, lens):
    """HessianAccumulator.flush_many -> llmc_hessian_accum_multi_*: up to four Hessians of one width share ONE unit queue (the
    three K = 4096 inputs of a Llama block). The token-chunk count of a launch is chosen for the problems that share it, so
    against one-by-one launches the fp32 sums are formed in another order: equal to summation-order noise, diagonal at the
    fp32 rounding of the exact value either way; the one-launch-per-problem form of the SAME call (k1_batch_off) is bit-identical."""
    from llmc_amd import _ffi
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    gen = torch.Generator(device='cuda').manual_seed(K)
    sets = [[(torch.randn(1, t, K, generator=gen, device='cuda') * (1 + p)).to(torch.bfloat16) for t in lens] for p in range(3)]
    singles = []
    for xs in sets:
        a = HessianAccumulator(K, 'cuda')
        for x in xs:
            a.add(x)
        singles.append(a.H.clone())
    merged = {}
    for batch_off in (0, 1):
        accs = [HessianAccumulator(K, 'cuda') for _ in sets]
        for a, xs in zip(accs, sets):
            for x in xs:
                a.add(x)
        accs[0].timing = []
        with _ffi.option(k1_batch_off=batch_off):
            HessianAccumulator.flush_many(accs)
        assert len(accs[0].timing) == 1 and accs[0].timing[0][4] == 3          # one call carried all three
        merged[batch_off] = [a.H.clone() for a in accs]
        for a, h, xs in zip(accs, singles, sets):
            assert not a._pending
            d = torch.sqrt(torch.outer(torch.diagonal(h), torch.diagonal(h)))
            assert float(((a.H - h).abs() / d).max()) < 2e-6, (K, batch_off)
            assert torch.equal(a.H, a.H.T)
            ref = sum((x.double() ** 2).sum(dim=(0, 1)) for x in xs) * (2.0 / len(xs))
            assert float(((torch.diagonal(a.H).double() - ref).abs() / ref).max()) <= 2.5e-7
        assert accs[0].barrier_timeouts() == 0
    for h0, h1 in zip(merged[0], merged[1]):
        assert torch.equal(h0, h1)
    # a second round on the same accumulators: the running mean continues through the merged launch
    for a, xs in zip(accs, sets):
        a.add(xs[0])
    HessianAccumulator.flush_many(accs)
    for a, xs in zip(accs, sets):
        ref = HessianAccumulator(K, 'cuda')
        for x in xs:
            ref.add(x)
        _ = ref.H
        ref.add(xs[0])
        d = torch.sqrt(torch.outer(torch.diagonal(ref.H), torch.diagonal(ref.H)))
        assert float(((a.H - ref.H).abs() / d).max()) < 2e-6


def test_hessians_of_different_widths_in_one_launch():
    """flush_many(..., mix_widths=True): problems of different K share one unit queue (the widest first); each H equals its
    one-by-one result up to the chunking's summation order, the diagonal at the fp32 rounding of the exact value."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    gen = torch.Generator(device='cuda').manual_seed(77)
    shapes = [(768, [2048, 2048, 1000]), (1280, [2048] * 4), (768, [512, 2048]), (256, [4096])]
    sets = [[(torch.randn(1, t, K, generator=gen, device='cuda') * (1 + i)).to(torch.bfloat16) for t in lens] for i, (K, lens) in enumerate(shapes)]
    accs = [HessianAccumulator(K, 'cuda') for K, _ in shapes]
    for a, xs in zip(accs, sets):
        for x in xs:
            a.add(x)
    accs[0].timing = accs[1].timing = []
    HessianAccumulator.flush_many(accs, mix_widths=True)
    assert len(accs[1].timing) == 1 and accs[1].timing[0][4] == 4          # one launch pair, four problems, the widest is its owner
    for a, xs in zip(accs, sets):
        ref = HessianAccumulator(a.K, 'cuda')
        for x in xs:
            ref.add(x)
        d = torch.sqrt(torch.outer(torch.diagonal(ref.H), torch.diagonal(ref.H)))
        assert float(((a.H - ref.H).abs() / d).max()) < 2e-6 and torch.equal(a.H, a.H.T)
        ex = sum((x.double() ** 2).sum(dim=(0, 1)) for x in xs) * (2.0 / len(xs))
        assert float(((torch.diagonal(a.H).double() - ex).abs() / ex).max()) <= 2.5e-7
