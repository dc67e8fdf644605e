"""Parity at the shapes bench.py actually times ("bench-path code with no parity evidence"):
  * K4 column loop with R >= 16384 rows (the 1024-thread k_gptq_block variant the gate|up stack uses),
  * K1 Hessian at T = 262144 / K = 4096 (the bench launch), K = 14336, a ragged K, f16, and a strided X above 4 GiB,
    checked against fp64 on sampled 256x256 tiles (incl. diagonal and last / ragged ones)."""
import numpy as np
import pytest
import torch

from oracle import gptq_ref as G
from oracle import quant_ref as Q

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


@pytest.mark.parametrize('R', [16384, 28672, 57344])      # 57344 = the Llama-3-70B gate|up stack
@pytest.mark.parametrize('static_groups', [False, True])
def test_column_loop_many_rows_bit_exact(R, static_groups):
    """R >= 16384 selects k_gptq_block<*, 1024> (gptq_loop.hip); bench's gate|up stack has R = 28672."""
    from llmc_amd.compression.quantization.gptq_ops import gptq_quantize
    K, bit, sym, gs = 512, 4, False, 128
    gen = torch.Generator().manual_seed(R + int(static_groups))
    W = (torch.randn(R, K, generator=gen) * 0.02).numpy()
    X = torch.randn(2 * K, K, generator=gen).double()
    H = (X.T @ X / K + 0.01 * torch.eye(K, dtype=torch.float64)).numpy()
    U = np.linalg.cholesky(np.linalg.inv(H)).T.astype(np.float32).copy()
    qmin, qmax = Q.int_range(bit, sym)
    scales = zeros = col_group = None
    ng = K // gs
    if static_groups:
        s, z = Q.minmax_qparams(W.reshape(-1, gs), 'f32', sym, qmin, qmax)
        scales, zeros = s.reshape(R, ng), z.reshape(R, ng)
        col_group = (np.random.RandomState(1).permutation(K) // gs).astype(np.int32)
    ref = G.weight_transform(W, U, sym, qmin, qmax, gs, static_groups, col_group, scales, zeros)
    tmp, losses, s, z = gptq_quantize(
        cu(W), cu(U), sym, qmin, qmax, gs, static_groups,
        None if col_group is None else torch.from_numpy(col_group).cuda(),
        None if scales is None else cu(scales), None if zeros is None else cu(zeros))
    np.testing.assert_array_equal(bits(tmp.cpu().numpy()), bits(ref['tmp']))
    np.testing.assert_array_equal(bits(losses.cpu().numpy()), bits(ref['losses']))
    if not static_groups:
        np.testing.assert_array_equal(bits(s.cpu().numpy()), bits(ref['scales']))
        np.testing.assert_array_equal(z.cpu().numpy(), ref['zeros'])


def synth_x(T, K, dtype, seed, ld=None):
    """SURVEY §8d activations, generated on the device in slabs; returns a [T, K] view (row stride ld)."""
    g = torch.Generator(device='cuda').manual_seed(seed)
    c = torch.exp(0.5 * torch.randn(K, generator=g, device='cuda'))
    c[torch.randperm(K, generator=g, device='cuda')[:8]] *= 100.0
    ld = ld or K
    buf = torch.empty((T, ld), device='cuda', dtype=dtype)
    if ld > K:
        buf[:, K:] = 1000.0          # padding must never leak into H
    step = max(1, (1 << 27) // K)
    for i in range(0, T, step):
        n = min(step, T - i)
        buf[i:i + n, :K] = (torch.randn((n, K), generator=g, device='cuda') * c).to(dtype)
    return buf[:, :K]


def check_tiles(H, x, n_seq, tiles, TM=256):
    """H tiles vs fp64 X_i^T X_j * 2/n_seq; error scale from torch's own fp32 GEMM of the same tile."""
    K = H.shape[0]
    worst = 0.0
    for bi, bj in tiles:
        ri = slice(bi * TM, min(K, (bi + 1) * TM))
        rj = slice(bj * TM, min(K, (bj + 1) * TM))
        xi, xj = x[:, ri], x[:, rj]
        ref = torch.zeros((ri.stop - ri.start, rj.stop - rj.start), dtype=torch.float64, device='cuda')
        ref32 = torch.zeros_like(ref, dtype=torch.float32)
        step = 32768
        for t in range(0, x.shape[0], step):
            a, b = xi[t:t + step], xj[t:t + step]
            ref += a.double().T @ b.double()
            ref32 += a.float().T @ b.float()
        ref *= 2.0 / n_seq
        ref32 *= 2.0 / n_seq
        di = torch.sqrt(torch.clamp((xi.double() ** 2).sum(0) * (2.0 / n_seq), min=1e-30))
        dj = torch.sqrt(torch.clamp((xj.double() ** 2).sum(0) * (2.0 / n_seq), min=1e-30))
        d = torch.outer(di, dj)
        e_ours = ((H[ri, rj].double() - ref).abs() / d).max().item()
        e_t = ((ref32.double() - ref).abs() / d).max().item()
        assert e_ours <= max(4 * e_t, 2e-6), ((bi, bj), e_ours, e_t)
        # mirrored tile is bit-identical
        assert torch.equal(H[ri, rj], H[rj, ri].T), (bi, bj)
        worst = max(worst, e_ours)
    return worst


def run_hessian(x, n_seq):
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    T, K = x.shape
    acc = HessianAccumulator(K, 'cuda')
    acc.add(x.as_strided((n_seq, T // n_seq, K), (x.stride(0) * (T // n_seq), x.stride(0), 1)))
    assert acc.nsamples == n_seq
    return acc.H


def test_hessian_bench_launch_k4096():
    """Exactly the bench launch for q|k|v, o, gate|up: 128 x 2048 tokens, K = 4096, bf16 (S = 9 token chunks)."""
    T, K, n_seq = 262144, 4096, 128
    x = synth_x(T, K, torch.bfloat16, 1)
    H = run_hessian(x, n_seq)
    tiles = [(0, 0), (15, 15), (7, 7), (15, 0), (9, 3), (12, 11), (1, 0), (15, 14), (8, 0), (5, 4)]
    check_tiles(H, x, n_seq, tiles)
    assert torch.isfinite(H).all()
    assert torch.equal(H, H.T)


def test_hessian_down_proj_k14336():
    """down_proj's input width (6384 padded tiles, X > 0.9 GiB per 32768 tokens); sampled tiles vs fp64."""
    T, K, n_seq = 32768, 14336, 16
    x = synth_x(T, K, torch.bfloat16, 2)
    H = run_hessian(x, n_seq)
    tiles = [(0, 0), (55, 55), (55, 0), (55, 54), (28, 27), (31, 4), (40, 40), (17, 16), (44, 9)]
    check_tiles(H, x, n_seq, tiles)
    assert torch.equal(H, H.T)


def test_hessian_down_proj_k14336_full_calibration_set():
    """Exactly the bench launch for down_proj: 128 x 2048 tokens, K = 14336, bf16 — 7 GiB of activations, S = 4
    token chunks of 65 536 tokens each, 1596 tiles; sampled tiles (diagonal ones included: their upper-right quadrant is the
    mirrored one) vs fp64, and the whole diagonal at the fp32 rounding of the exact sum of squares."""
    T, K, n_seq = 262144, 14336, 128
    x = synth_x(T, K, torch.bfloat16, 7)
    H = run_hessian(x, n_seq)
    tiles = [(0, 0), (55, 55), (55, 0), (55, 54), (28, 27), (31, 4), (40, 40), (17, 16), (44, 9), (27, 27)]
    check_tiles(H, x, n_seq, tiles)
    assert torch.equal(H, H.T) and torch.isfinite(H).all()
    d = torch.zeros(K, dtype=torch.float64, device='cuda')
    for t in range(0, T, 16384):
        d += (x[t:t + 16384].double() ** 2).sum(0)
    d *= 2.0 / n_seq
    # an fp32 rounding of the exact value is 6e-8; the fp32 chains of 128 MFMAs between the fp64 folds add ~1e-7 (the fp32 chain over a
    # whole chunk left 3.2e-6 here, the reference's own sgemm 1.2-1.7e-6)
    assert float(((torch.diagonal(H).double() - d).abs() / d).max()) <= 3e-7


def test_hessian_ragged_k_and_f16():
    T, K, n_seq = 65536, 5000, 32       # 5000 = 19 * 256 + 136: ragged last tile row / column
    x = synth_x(T, K, torch.float16, 3)
    x.mul_(0.05)                          # keep 100x outlier channels inside fp16 range
    H = run_hessian(x, n_seq)
    tiles = [(19, 19), (19, 0), (19, 18), (0, 0), (10, 10), (12, 5), (18, 17), (4, 3)]
    check_tiles(H, x, n_seq, tiles)
    assert torch.equal(H, H.T)


def test_hessian_strided_rows_above_4gib():
    """ldx > K with T * ldx * 2 > 4 GiB: chunk bases keep every 32-bit buffer offset below 2^32."""
    T, K, ld, n_seq = 262144, 4096, 8448, 128      # 262144 * 8448 * 2 = 4.43e9 bytes
    x = synth_x(T, K, torch.bfloat16, 4, ld=ld)
    assert x.stride(0) == ld and T * ld * 2 > (1 << 32)
    H = run_hessian(x, n_seq)
    tiles = [(0, 0), (15, 15), (15, 0), (8, 7), (3, 2), (11, 6), (14, 13), (6, 6)]
    check_tiles(H, x, n_seq, tiles)


def test_hessian_per_sample_table_bit_identical_at_the_bench_launch():
    """The reference's calling pattern (calib.bs = 1: 128 hook calls of [1, 2048, K], gptq_w_only.yml:12): the deferred
    per-sample path walks 128 separately allocated tensors through the sample table in ONE launch and gives the bits
    of the one-tensor launch over the same tokens."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    T, K, n_seq = 262144, 4096, 128
    x = synth_x(T, K, torch.bfloat16, 6)
    H1 = run_hessian(x, n_seq).clone()
    seq = T // n_seq
    samples = [x[i * seq:(i + 1) * seq].clone().unsqueeze(0) for i in range(n_seq)]    # separate allocations
    acc = HessianAccumulator(K, 'cuda')
    acc.timing = []
    for smp in samples:
        acc.add(smp)
    H2 = acc.H
    assert acc.nsamples == n_seq and len(acc.timing) == 1          # one launch for the 128 calls
    assert torch.equal(H1, H2)
