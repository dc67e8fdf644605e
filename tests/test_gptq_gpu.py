"""K2/K3/K4 on MI355X vs the oracle and the reference's golden vectors."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden
from llmc_amd import _ffi
from oracle import gptq_ref as G
from oracle import quant_ref as Q

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def sgemm(A, B, C, M, N, Kd, TA, TB, epi, hints=(0, 0, 0, 0)):
    L = _ffi.lib()
    _ffi.check(L.llmc_test_sgemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), A.stride(0), B.stride(0), C.stride(0),
                                 M, N, Kd, int(TA), int(TB), epi, *hints, _ffi.stream()), 'sgemm')
    return C


# Kd <= 128 runs the short-K kernel, larger Kd the 128x128-tile kernel
@pytest.mark.parametrize('shape', [(96, 200, 128), (300, 260, 128), (128, 128, 16), (257, 513, 72), (1024, 3968, 128),
                                   (200, 300, 512), (257, 129, 200), (64, 64, 2), (130, 70, 127)])
def test_sgemm_bitwise_chain(shape):
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, Kd, generator=gen) * 0.01)
    b = torch.randn(Kd, (N + 3) // 4 * 4, generator=gen)[:, :N]
    ref = G.mm_chain(a.numpy(), b.numpy())
    ldn = (N + 3) // 4 * 4
    bd = torch.zeros(Kd, ldn).cuda()
    bd[:, :N] = b.cuda()
    # NN: A stored [M x Kd]
    Kp = (Kd + 3) // 4 * 4
    ad = torch.zeros(M, Kp).cuda()
    ad[:, :Kd] = a.cuda()
    c0 = torch.randn(M, ldn, generator=gen)
    C = c0.cuda()
    sgemm(ad, bd, C, M, N, Kd, False, False, 0)
    exp = c0.numpy()[:, :N] - ref
    np.testing.assert_array_equal(bits(C.cpu().numpy()[:, :N]), bits(exp))
    # TN: A stored [Kd x M]
    Mp = (M + 3) // 4 * 4
    at = torch.zeros(Kd, Mp).cuda()
    at[:, :M] = a.t().cuda()
    C2 = torch.zeros(M, ldn).cuda()
    sgemm(at, bd, C2, M, N, Kd, True, False, 1)
    np.testing.assert_array_equal(bits(C2.cpu().numpy()[:, :N]), bits(ref))
    C3 = torch.zeros(M, ldn).cuda()
    sgemm(at, bd, C3, M, N, Kd, True, False, 2)
    np.testing.assert_array_equal(bits(C3.cpu().numpy()[:, :N]), bits(-ref))


def sgemm_phased(A, B, C, M, N, Kd, TA, phase):
    L = _ffi.lib()
    _ffi.check(L.llmc_test_sgemm_phased(A.data_ptr(), B.data_ptr(), C.data_ptr(), A.stride(0), B.stride(0), C.stride(0),
                                        M, N, Kd, int(TA), phase, _ffi.stream()), 'sgemm_phased')
    return C


@pytest.mark.parametrize('shape', [(256, 128, 128), (512, 384, 512), (1024, 1152, 256), (4096, 2176, 512), (768, 4224, 384), (384, 256, 256)])
def test_sgemm_wide_phased_far_update_bitwise(shape):
    """K4's lazy far update (gptq.py:240-244 for a group of 128-column blocks in one launch) on k_sgemm_wide (256 x 128 tiles,
    LDS-DMA ring, one wave per SIMD): bit-identical to the oracle's chain applied block by block, and to k_sgemm."""
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(M + N + Kd)
    at = (torch.randn(Kd, M, generator=gen) * 0.01)          # the error columns, k-major
    b = torch.randn(Kd, N, generator=gen)
    c0 = torch.randn(M, N, generator=gen)
    exp = c0.numpy().copy()
    if M * N * Kd <= 512 * 384 * 512:                        # the oracle's scalar chain is slow: small cases only
        for p0 in range(0, Kd, 128):
            exp = exp - G.mm_chain(np.ascontiguousarray(at[p0:p0 + 128].t().numpy()), b[p0:p0 + 128].numpy())
    else:
        exp = None
    res = {}
    for name, opts in (('wide4', dict(no_shortk=1, sgemm_no_wide=4)), ('wide2', dict(no_shortk=1, sgemm_no_wide=2)),
                       ('k_sgemm', dict(no_shortk=1, sgemm_no_wide=1)), ('default', {})):
        with _ffi.option(**opts):
            C = c0.cuda()
            sgemm_phased(at.cuda(), b.cuda(), C, M, N, Kd, True, 128)
            res[name] = C.cpu().numpy()
    if exp is not None:
        np.testing.assert_array_equal(bits(res['wide4']), bits(exp))
    for name in ('wide4', 'wide2', 'default'):
        np.testing.assert_array_equal(bits(res[name]), bits(res['k_sgemm']))


def test_sgemm_wide_strided_operands_and_zero_products():
    """Operands that are column ranges of wider matrices (ld > width, as in the column loop), and exact zeros in the error panel
    (a phase whose products are all zero must leave -0.0 entries of C as they are: C - (+0))."""
    M, N, Kd = 512, 640, 256
    gen = torch.Generator().manual_seed(5)
    Abig = (torch.randn(Kd, M + 64, generator=gen) * 0.01).cuda()
    Abig[:128] = 0.0
    Bbig = torch.randn(Kd, N + 256, generator=gen).cuda()
    Cbig = torch.randn(M, N + 256, generator=gen)
    Cbig[::7, ::5] = -0.0
    out = {}
    for name, opts in (('wide4', dict(no_shortk=1, sgemm_no_wide=4)), ('wide2', dict(no_shortk=1, sgemm_no_wide=2)),
                       ('k_sgemm', dict(no_shortk=1, sgemm_no_wide=1))):
        with _ffi.option(**opts):
            C = Cbig.cuda()
            sgemm_phased(Abig[:, 64:], Bbig[:, 128:], C[:, 128:], M, N, Kd, True, 128)
            out[name] = C.cpu().numpy()
    for name in ('wide4', 'wide2'):
        np.testing.assert_array_equal(bits(out[name]), bits(out['k_sgemm']))
        np.testing.assert_array_equal(bits(out[name][:, :128]), bits(Cbig.numpy()[:, :128]))      # columns outside the product untouched


def gemm3(A, B, C, M, N, Kd, TA, epi, hints=(0, 0, 0)):
    L = _ffi.lib()
    _ffi.check(L.llmc_test_gemm3(A.data_ptr(), B.data_ptr(), C.data_ptr(), A.stride(0), B.stride(0), C.stride(0),
                                 M, N, Kd, int(TA), epi, *hints, _ffi.stream()), 'gemm3')
    return C


@pytest.mark.parametrize('shape', [(128, 128, 32), (384, 512, 512), (1000, 776, 200), (2048, 2048, 512)])
@pytest.mark.parametrize('TA', [True, False])
def test_gemm3_split_bf16_matches_fp32_accuracy(shape, TA):
    """C (op) op(A) B on the 16-bit pipe with 3-term bf16 operands: error vs fp64 within 2x of the exact fp32 chain's."""
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(M + Kd)
    At = (torch.randn(Kd, M, generator=gen) * torch.exp(torch.randn(Kd, 1, generator=gen)))
    B = (torch.randn(Kd, N, generator=gen) * torch.exp(torch.randn(Kd, 1, generator=gen))).cuda()
    Kp = (Kd + 3) // 4 * 4
    if TA:
        A = At.cuda()
    else:
        A = torch.zeros(M, Kp)
        A[:, :Kd] = At.T
        A = A.cuda()
    C0 = torch.randn(M, N, generator=gen).cuda()
    ref_prod = At.double().T.cuda() @ B.double()
    scale = (At.double().abs().T.cuda() @ B.double().abs()).max()
    for epi, ref in ((0, C0.double() - ref_prod), (1, ref_prod), (2, -ref_prod)):
        C3 = gemm3(A, B, C0.clone(), M, N, Kd, TA, epi)
        C1 = sgemm(A, B, C0.clone(), M, N, Kd, TA, False, epi)
        e3 = ((C3.double() - ref).abs().max() / scale).item()
        e1 = ((C1.double() - ref).abs().max() / scale).item()
        assert e3 <= max(2 * e1, 2e-7), (epi, e3, e1)
    if M == N and TA:   # upper tiles only
        C3 = gemm3(A, B, C0.clone(), M, N, Kd, TA, 0)
        Cu = gemm3(A, B, C0.clone(), M, N, Kd, TA, 0, (0, 0, 1))
        iu = torch.triu(torch.ones(M, N, dtype=torch.bool)).cuda()
        assert torch.equal(Cu[iu], C3[iu])


@pytest.mark.parametrize('shape', [(256, 128, 128), (512, 640, 512), (1000, 776, 192), (260, 132, 256), (1024, 1024, 512),
                                   (768, 768, 128)])
def test_gemm3_specialised_kernel_is_bit_identical(shape, monkeypatch):
    """k_gemm3s (producer waves split the panels, MFMA waves multiply; range-checked buffer loads two K-steps ahead) against
    k_gemm3 (option gemm3_nospec): same bits for every epilogue and for the upper-only form, on panels that sit inside a
    larger poisoned matrix (anything read beyond a panel's rows or columns would show), and nothing written outside C's
    upper part when only that is asked for."""
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(M * 7 + Kd)
    ld = 2304
    big = torch.full((Kd + 64, ld), float('nan'))
    big[:Kd, 8:8 + M] = torch.randn(Kd, M, generator=gen) * torch.exp(torch.randn(Kd, 1, generator=gen))
    big[:Kd, 1200:1200 + N] = torch.randn(Kd, N, generator=gen)
    big = big.cuda()
    A = big[:, 8:]
    B = big[:, 1200:]
    C0 = torch.randn(M, N, generator=gen).cuda()
    _ffi.set_option('gemm3s_min_tiles', 1)
    for epi in (0, 1, 2):
        for hints in ((0, 0, 0), (0, 0, 1)) if M == N else ((0, 0, 0),):
            _ffi.set_option('gemm3_nospec', 0)
            c_new = gemm3(A, B, C0.clone(), M, N, Kd, True, epi, hints)
            _ffi.set_option('gemm3_nospec', 1)
            c_old = gemm3(A, B, C0.clone(), M, N, Kd, True, epi, hints)
            if hints[2]:
                iu = torch.triu(torch.ones(M, N, dtype=torch.bool)).cuda()
                assert torch.equal(c_new[iu], c_old[iu]), (epi, hints)
                # 128 x 128 blocks strictly below the diagonal blocks are untouched by both
                blk = (torch.arange(M)[:, None] // 128 > torch.arange(N)[None, :] // 128).cuda()
                assert torch.equal(c_new[blk], C0[blk])
            else:
                assert not torch.isnan(c_new).any()
                assert torch.equal(c_new, c_old), (epi, hints)


@pytest.mark.parametrize('shape', [(256, 128, 128), (512, 640, 512), (1000, 776, 192), (264, 136, 256), (1024, 1024, 512)])
def test_gemm3_planes_form_is_bit_identical(shape, monkeypatch):
    """The far updates' form with the panels split once into bf16 planes in memory (k_split3_planes + k_gemm3s copying
    planes) against k_gemm3 splitting inside every tile: same bits, every epilogue, upper-only included."""
    L = _ffi.lib()
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(M * 5 + Kd)
    ld = 2304
    big = torch.full((Kd + 64, ld), float('nan'))
    big[:Kd, 8:8 + M] = torch.randn(Kd, M, generator=gen) * torch.exp(torch.randn(Kd, 1, generator=gen))
    big[:Kd, 1200:1200 + N] = torch.randn(Kd, N, generator=gen)
    big = big.cuda()
    A = big[:, 8:]
    B = big[:, 1200:]
    C0 = torch.randn(M, N, generator=gen).cuda()
    ldp = (max(M, N) + 7) // 8 * 8
    ws = torch.full((6 * Kd * ldp,), -1, dtype=torch.int16).cuda()
    _ffi.set_option('gemm3s_min_tiles', 1)
    for epi in (0, 1, 2):
        for upper in (0, 1) if M == N else (0,):
            _ffi.set_option('gemm3_nospec', 0)
            c_new = C0.clone()
            _ffi.check(L.llmc_test_gemm3_planes(A.data_ptr(), B.data_ptr(), c_new.data_ptr(), A.stride(0), B.stride(0), c_new.stride(0),
                                                M, N, Kd, epi, upper, ws.data_ptr(), _ffi.stream()), 'gemm3 planes')
            _ffi.set_option('gemm3_nospec', 1)
            c_old = gemm3(A, B, C0.clone(), M, N, Kd, True, epi, (0, 0, upper))
            if upper:
                iu = torch.triu(torch.ones(M, N, dtype=torch.bool)).cuda()
                assert torch.equal(c_new[iu], c_old[iu]), (epi, upper)
                blk = (torch.arange(M)[:, None] // 128 > torch.arange(N)[None, :] // 128).cuda()
                assert torch.equal(c_new[blk], C0[blk])
            else:
                assert not torch.isnan(c_new).any()
                assert torch.equal(c_new, c_old), (epi, upper)


@pytest.mark.parametrize('shape', [(512, 640, 512), (1024, 1024, 512), (2048, 2048, 256), (1536, 2560, 128), (128, 128, 128)])
def test_gemm3w_two_workgroups_per_cu_is_bit_identical(shape):
    """K3's far update on k_gemm3w (128 x 128 tiles, two workgroups per CU, LDS-DMA ring of three stages, fragments a stage ahead)
    against k_gemm3s (option gemm3_no_wide) and k_gemm3 (gemm3_nospec): the same bits, full and upper-only."""
    L = _ffi.lib()
    M, N, Kd = shape
    gen = torch.Generator().manual_seed(M * 3 + N + Kd)
    A = (torch.randn(Kd, M, generator=gen) * torch.exp(torch.randn(Kd, 1, generator=gen))).cuda()
    B = torch.randn(Kd, N, generator=gen).cuda()
    Cbig = torch.randn(M, N + 64, generator=gen).cuda()          # C is a column range of a wider matrix
    ldp = (max(M, N) + 7) // 8 * 8
    ws = torch.full((6 * Kd * ldp,), -1, dtype=torch.int16).cuda()
    _ffi.set_option('gemm3s_min_tiles', 1)
    for upper in (0, 1) if M == N else (0,):
        out = {}
        for name, opts in (('w', {}), ('s', dict(gemm3_no_wide=1))):
            with _ffi.option(**opts):
                c = Cbig.clone()
                _ffi.check(L.llmc_test_gemm3_planes(A.data_ptr(), B.data_ptr(), c[:, 64:].data_ptr(), A.stride(0), B.stride(0), c.stride(0),
                                                    M, N, Kd, 0, upper, ws.data_ptr(), _ffi.stream()), 'gemm3 planes')
                out[name] = c
        with _ffi.option(gemm3_nospec=1):
            ref = Cbig.clone()
            gemm3(A, B, ref[:, 64:], M, N, Kd, True, 0, (0, 0, upper))
        assert torch.equal(out['w'][:, :64], Cbig[:, :64])
        if upper:
            iu = torch.triu(torch.ones(M, N, dtype=torch.bool)).cuda()
            for k in ('w', 's'):
                assert torch.equal(out[k][:, 64:][iu], ref[:, 64:][iu]), (k, upper)
            blk = (torch.arange(M)[:, None] // 128 > torch.arange(N)[None, :] // 128).cuda()
            assert torch.equal(out['w'][:, 64:][blk], Cbig[:, 64:][blk])
        else:
            assert torch.equal(out['w'], ref) and torch.equal(out['s'], ref)


def test_gemm3_triangular_hints_do_not_change_results():
    n = 384
    gen = torch.Generator().manual_seed(3)
    A = torch.triu(torch.randn(n, n, generator=gen)).cuda()
    B = torch.triu(torch.randn(n, n, generator=gen)).cuda()
    X = torch.randn(n, n, generator=gen).cuda()
    full = gemm3(A, X, torch.zeros(n, n).cuda(), n, n, n, False, 1)
    hint = gemm3(A, X, torch.zeros(n, n).cuda(), n, n, n, False, 1, (1, 0, 0))
    assert torch.equal(full, hint)
    full = gemm3(X, B, torch.zeros(n, n).cuda(), n, n, n, False, 2)
    hint = gemm3(X, B, torch.zeros(n, n).cuda(), n, n, n, False, 2, (0, 1, 0))
    assert torch.equal(full, hint)


def test_sgemm_triangular_hints_do_not_change_bits():
    n = 384
    gen = torch.Generator().manual_seed(1)
    A = torch.triu(torch.randn(n, n, generator=gen))
    B = torch.triu(torch.randn(n, n, generator=gen))
    X = torch.randn(n, n, generator=gen)
    Ad, Bd, Xd = A.cuda(), B.cuda(), X.cuda()
    full = sgemm(Ad, Xd, torch.zeros(n, n).cuda(), n, n, n, False, False, 1)
    hint = sgemm(Ad, Xd, torch.zeros(n, n).cuda(), n, n, n, False, False, 1, (1, 0, 0, 0))
    assert torch.equal(full, hint)
    full = sgemm(Xd, Bd, torch.zeros(n, n).cuda(), n, n, n, False, False, 2)
    hint = sgemm(Xd, Bd, torch.zeros(n, n).cuda(), n, n, n, False, False, 2, (0, 0, 1, 0))
    assert torch.equal(full, hint)
    # lower-triangular op(A) = V^T with V upper stored k-major
    full = sgemm(Ad, Xd, torch.zeros(n, n).cuda(), n, n, n, True, False, 1)
    hint = sgemm(Ad, Xd, torch.zeros(n, n).cuda(), n, n, n, True, False, 1, (0, 1, 0, 0))
    assert torch.equal(full, hint)


def test_sgemm_shortk_hints_do_not_change_bits():
    n = 128
    gen = torch.Generator().manual_seed(2)
    V = torch.triu(torch.randn(n, n, generator=gen))
    X = torch.randn(n, 520, generator=gen)
    Vd, Xd = V.cuda(), X.cuda()
    # panel solve shape: op(A) = V^T (lower), Kd = 128
    full = sgemm(Vd, Xd, torch.zeros(n, 520).cuda(), n, 520, n, True, False, 1)
    hint = sgemm(Vd, Xd, torch.zeros(n, 520).cuda(), n, 520, n, True, False, 1, (0, 1, 0, 0))
    assert torch.equal(full, hint)
    # the same product in place (C aliases B), as the factorisation's panel solve runs it: one workgroup per column
    # block walks the row tiles, so no tile can read rows another one has already overwritten
    for n_cols in (520, 8192):
        Xb = torch.randn(n, n_cols, generator=gen).cuda()
        ref = sgemm(Vd, Xb, torch.zeros(n, n_cols).cuda(), n, n_cols, n, True, False, 1, (0, 1, 0, 0))
        inp = Xb.clone()
        sgemm(Vd, inp, inp, n, n_cols, n, True, False, 1, (0, 1, 0, 0))
        assert torch.equal(inp, ref)
    # symmetric update, upper tiles only: tiles that touch j >= i equal the full result, the others are untouched
    P = torch.randn(n, 384, generator=gen).cuda()
    C0 = torch.randn(384, 384, generator=gen).cuda()
    full = sgemm(P, P, C0.clone(), 384, 384, n, True, False, 0)
    up = sgemm(P, P, C0.clone(), 384, 384, n, True, False, 0, (0, 0, 0, 1))
    iu = torch.triu(torch.ones(384, 384, dtype=torch.bool)).cuda()
    assert torch.equal(full[iu], up[iu])
    tile_lower = (torch.arange(384)[None, :] // 64 * 64 + 64 <= torch.arange(384)[:, None] // 64 * 64).cuda()
    assert torch.equal(up[tile_lower], C0[tile_lower])


def test_column_loop_bit_exact_vs_reference_golden():
    from llmc_amd.compression.quantization.gptq_ops import gptq_quantize
    g = load_golden('gptq+more')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        bit, sym, gs, actorder, static_groups, R, K, qmin, qmax = g[p + 'meta']
        sym, gs, static_groups, R, K = bool(sym), int(gs), bool(static_groups), int(R), int(K)
        perm = g[p + 'perm']
        scales = zeros = col_group = None
        if static_groups or gs == 0:
            ng = 1 if gs == 0 else K // gs
            scales = cu(g[p + 'buf_scales'].reshape(R, ng))
            zeros = cu(g[p + 'buf_zeros'].reshape(R, ng)) if g[p + 'buf_zeros'].size else None
            if gs:
                idx = perm if perm.size else np.arange(K)
                col_group = torch.from_numpy((idx // gs).astype(np.int32)).cuda()
        tmp, losses, s, z = gptq_quantize(cu(g[p + 'Wp']), cu(g[p + 'U']), sym, qmin, qmax, gs, static_groups,
                                          col_group, scales, zeros)
        np.testing.assert_array_equal(bits(tmp.cpu().numpy()), bits(g[p + 'tmp']), err_msg=name)
        np.testing.assert_array_equal(bits(losses.cpu().numpy()), bits(g[p + 'losses']), err_msg=name)
        if not static_groups and gs:
            np.testing.assert_array_equal(bits(s.cpu().numpy()), bits(g[p + 'g_scales']), err_msg=name)
            if not sym:
                np.testing.assert_array_equal(z.cpu().numpy(), g[p + 'g_zeros'], err_msg=name)


@pytest.mark.parametrize('cfg', [(1024, 1024, 4, False, 128, False), (512, 2048, 4, True, 128, True),
                                 (333, 640, 4, False, 64, False), (256, 512, 8, True, 0, False),
                                 (200, 1000, 4, True, 0, False), (64, 2304, 4, False, 128, False)])
def test_column_loop_bit_exact_vs_oracle_larger(cfg):
    from llmc_amd.compression.quantization.gptq_ops import gptq_quantize
    R, K, bit, sym, gs, static_groups = cfg
    gen = torch.Generator().manual_seed(R + K)
    W = (torch.randn(R, K, generator=gen) * 0.02).numpy()
    X = torch.randn(2 * K, K, generator=gen).double()
    H = (X.T @ X / K + 0.01 * torch.eye(K, dtype=torch.float64)).numpy()
    Hinv = np.linalg.inv(H)
    U = np.linalg.cholesky(Hinv).T.astype(np.float32).copy()
    qmin, qmax = Q.int_range(bit, sym)
    scales = zeros = col_group = None
    ng = 1 if gs == 0 else K // gs
    if static_groups or gs == 0:
        w2 = W.reshape(-1, gs if gs else K)
        s, z = Q.minmax_qparams(w2, 'f32', sym, qmin, qmax)
        scales, zeros = s.reshape(R, ng), (None if sym else z.reshape(R, ng))
        if gs:
            perm = np.random.RandomState(0).permutation(K)
            col_group = (perm // gs).astype(np.int32)
    ref = G.weight_transform(W, U, sym, qmin, qmax, gs, static_groups, col_group, scales, zeros)
    tmp, losses, s, z = gptq_quantize(
        cu(W), cu(U), sym, qmin, qmax, gs, static_groups,
        None if col_group is None else torch.from_numpy(col_group).cuda(),
        None if scales is None else cu(scales), None if zeros is None else cu(zeros))
    np.testing.assert_array_equal(bits(tmp.cpu().numpy()), bits(ref['tmp']))
    np.testing.assert_array_equal(bits(losses.cpu().numpy()), bits(ref['losses']))
    if not static_groups and gs:
        np.testing.assert_array_equal(bits(s.cpu().numpy()), bits(ref['scales']))
        np.testing.assert_array_equal(z.cpu().numpy(), ref['zeros'])


@pytest.mark.parametrize('static_groups', [False, True])
def test_column_loop_non_plain_operands_fall_back_bit_exact(static_groups):
    """Operands outside the fast in-block path's "plain" range (-0, denormal, tiny, huge, all-zero rows) make the
    owning wave redo the block with the generic IEEE-division path: results stay bit-identical to the oracle."""
    from llmc_amd.compression.quantization.gptq_ops import gptq_quantize
    R, K, bit, sym, gs = 256, 512, 4, False, 128
    gen = torch.Generator().manual_seed(77)
    W = (torch.randn(R, K, generator=gen) * 0.02).numpy()
    W[0, 5] = -0.0
    W[1, 130:140] = -0.0
    W[8, 7] = 1e-42          # denormal
    W[9, 300] = 1e-30        # below 2^-40
    W[16, 64] = 1e13         # above 2^40
    W[24, :] = 0.0           # scale clamps to 1e-5, every diff is +0
    W[32, 128:256] = 0.0
    W[40, 200] = np.float32(0.02) * 0 - 0.0
    X = torch.randn(2 * K, K, generator=gen).double()
    H = (X.T @ X / K + 0.01 * torch.eye(K, dtype=torch.float64)).numpy()
    U = np.linalg.cholesky(np.linalg.inv(H)).T.astype(np.float32).copy()
    qmin, qmax = Q.int_range(bit, sym)
    scales = zeros = col_group = None
    ng = K // gs
    if static_groups:
        s, z = Q.minmax_qparams(W.reshape(-1, gs), 'f32', sym, qmin, qmax)
        scales, zeros = s.reshape(R, ng), z.reshape(R, ng)
        col_group = (np.arange(K) // gs).astype(np.int32)
    ref = G.weight_transform(W, U, sym, qmin, qmax, gs, static_groups, col_group, scales, zeros)
    tmp, losses, s, z = gptq_quantize(
        cu(W), cu(U), sym, qmin, qmax, gs, static_groups,
        None if col_group is None else torch.from_numpy(col_group).cuda(),
        None if scales is None else cu(scales), None if zeros is None else cu(zeros))
    np.testing.assert_array_equal(bits(tmp.cpu().numpy()), bits(ref['tmp']))
    np.testing.assert_array_equal(bits(losses.cpu().numpy()), bits(ref['losses']))


@pytest.mark.parametrize('shape', [(37, 128), (300, 4096), (16, 14336)])
def test_gather_cols_matches_index_select(shape):
    from llmc_amd.compression.quantization.gptq_ops import gather_cols
    R, K = shape
    gen = torch.Generator().manual_seed(K)
    src = torch.randn(R, K, generator=gen).cuda()
    idx = torch.randperm(K, generator=gen).cuda()
    assert torch.equal(gather_cols(src, idx), src.index_select(1, idx))


@pytest.mark.parametrize('wdtype', [torch.float16, torch.bfloat16, torch.float32])
def test_hessian_prep_lds_gather_matches_definition(wdtype):
    from llmc_amd.compression.quantization.gptq_ops import hessian_prep
    K, R = 1024, 77
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(2 * K, K, generator=gen)
    H = (X.T @ X / K).cuda()
    H[:, 17] = 0
    H[17, :] = 0
    H[:, 900] = 0
    H[900, :] = 0
    W = torch.randn(R, K, generator=gen).to(wdtype).cuda()
    perm = torch.randperm(K, generator=gen).cuda()
    Hd = H.clone()
    Hp, Wp = hessian_prep(Hd, W, perm, 0.01)
    Href = H.clone()
    dead = torch.diagonal(Href) == 0
    Href[dead, dead] = 1
    damp = (0.01 * torch.diagonal(Href).double().mean()).float()
    Wref = W.float().clone()
    Wref[:, dead] = 0
    Href = Href[perm][:, perm] + torch.eye(K, device='cuda') * damp
    assert torch.equal(Wp, Wref[:, perm])
    assert torch.equal(torch.diagonal(Hd), torch.diagonal(torch.where(torch.eye(K, device='cuda', dtype=torch.bool), torch.where(H == 0, torch.ones_like(H), H), H)))
    off = ~torch.eye(K, dtype=torch.bool, device='cuda')
    assert torch.equal(Hp[off], Href[off])
    assert torch.allclose(torch.diagonal(Hp), torch.diagonal(Href), rtol=1e-6, atol=0)


def test_factor_and_column_loop_are_run_to_run_deterministic():
    """Helper-stream overlap must not change a single bit between runs (a race would): 4 runs each of the K = 4096
    factorisation and of a 2048 x 4096 column loop, which both use the look-ahead stream."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper, gptq_quantize
    K = 4096
    gen = torch.Generator(device='cuda').manual_seed(11)
    X = torch.randn(2 * K, K, generator=gen, device='cuda')
    H = X.T @ X / K
    H += 0.01 * torch.diagonal(H).mean() * torch.eye(K, device='cuda')
    Us = [chol_inv_upper(H.clone(), check=False).clone() for _ in range(4)]
    for u in Us[1:]:
        assert torch.equal(u, Us[0])
    W = torch.randn(2048, K, generator=gen, device='cuda') * 0.02
    outs = [gptq_quantize(W.clone(), Us[0], False, 0.0, 15.0, 128) for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


def test_split_bf16_factor_is_as_accurate_as_fp32_on_outlier_channels(monkeypatch):
    """K3's large products run as split-bf16 by default; on a Hessian with 100x outlier channels (condition number
    ~1e6 after damping) its factor must be as close to the fp64 factor as the all-fp32-MFMA path's (option k3_fp32)."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    K = 2048
    gen = torch.Generator().manual_seed(5)
    c = torch.exp(0.5 * torch.randn(K, generator=gen, dtype=torch.float64))
    c[torch.randperm(K, generator=gen)[:8]] *= 100
    X = torch.randn(3 * K, K, generator=gen, dtype=torch.float64) * c
    H = (X.T @ X) * (2.0 / 3)
    H = H[torch.argsort(torch.diagonal(H), descending=True)][:, torch.argsort(torch.diagonal(H), descending=True)]
    H += 0.01 * H.diag().mean() * torch.eye(K, dtype=torch.float64)
    Hd = H.float().cuda()
    Uref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd.double().cpu())), upper=True)
    _ffi.set_option('k3_fp32', 0)
    U3 = chol_inv_upper(Hd.clone()).double().cpu()
    _ffi.set_option('k3_fp32', 1)
    U1 = chol_inv_upper(Hd.clone()).double().cpu()
    e3 = ((U3 - Uref).abs().max() / Uref.abs().max()).item()
    e1 = ((U1 - Uref).abs().max() / Uref.abs().max()).item()
    assert not torch.equal(U3, U1)          # the two paths really are different arithmetic
    assert e3 <= max(2 * e1, 1e-6), (e3, e1)


@pytest.mark.parametrize('K', [128, 384, 1000, 4096, 6144])
def test_chol_inv_upper_vs_fp64(K):
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    gen = torch.Generator().manual_seed(K)
    X = torch.randn(3 * K, K, generator=gen, dtype=torch.float64)
    X[:, ::7] *= 5
    H = (X.T @ X) * (2.0 / 3)
    H += 0.01 * H.diag().mean() * torch.eye(K, dtype=torch.float64)
    Hd = H.float().cuda()
    U = chol_inv_upper(Hd.clone()).double().cpu()
    assert torch.equal(U, torch.triu(U))
    # defining property: U^T U = H^-1  <=>  U H U^T = I
    Hf = Hd.double().cpu()
    E = U @ Hf @ U.T - torch.eye(K, dtype=torch.float64)
    L = torch.linalg.cholesky(Hf)
    Uref = torch.linalg.cholesky(torch.cholesky_inverse(L), upper=True)
    # reference's own fp32 3-call route, for the error scale
    Lf = torch.linalg.cholesky(Hf.float())
    U32 = torch.linalg.cholesky(torch.cholesky_inverse(Lf), upper=True).double()
    e_ref = (U32 - Uref).abs().max() / Uref.abs().max()
    e_ours = (U - Uref).abs().max() / Uref.abs().max()
    assert e_ours <= max(4 * e_ref, 1e-5), (float(e_ours), float(e_ref))
    assert E.abs().max() < 5e-3


@pytest.mark.parametrize('K', [4096, 5000])
def test_chol_inv_upper_far_updates_on_planes_keep_every_bit(K, monkeypatch):
    """K3 with the large far updates on pre-split planes (the default) against the same factorisation with k_gemm3 splitting
    inside every tile (option k3_no_planes): the factor is the same to the last bit. K = 5000: far widths that are multiples
    of 8 but not of 128 (ragged last tiles)."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    gen = torch.Generator().manual_seed(K + 1)
    X = torch.randn(2 * K, K, generator=gen)
    X[:, ::5] *= 4
    H = ((X.T @ X) / K).cuda()
    H += 0.01 * H.diag().mean() * torch.eye(K, device='cuda')
    _ffi.set_option('k3_no_planes', 0)
    u_planes = chol_inv_upper(H.clone())
    _ffi.set_option('k3_no_planes', 1)
    u_plain = chol_inv_upper(H.clone())
    assert torch.equal(u_planes, u_plain)
    # the far update of an outer block as two launches (next block's rows, the rest) instead of one, and with the library's
    # helper streams (always two launches, the second beside the next block's chain)
    _ffi.set_option('k3_no_planes', 0)
    _ffi.set_option('k3_split_far', 1)
    assert torch.equal(chol_inv_upper(H.clone()), u_planes)
    _ffi.set_option('k3_split_far', 0)
    prev = _ffi.lib().llmc_hip_set_helper_streams(0)
    try:
        u_single = chol_inv_upper(H.clone())
        _ffi.lib().llmc_hip_set_helper_streams(1)
        u_helper = chol_inv_upper(H.clone())
    finally:
        _ffi.lib().llmc_hip_set_helper_streams(prev)
    assert torch.equal(u_single, u_planes) and torch.equal(u_helper, u_planes)


@pytest.mark.parametrize('R,K', [(1024, 2304), (384, 1536)])
def test_column_loop_same_bits_on_every_far_update_kernel(R, K):
    """The whole column loop (gptq.py:199-244) with its far updates on k_sgemm_wide's two forms and on k_sgemm: quantized weights,
    losses, scales and zeros identical to the last bit (R = 384: rows that are no multiple of 256 — the 128 x 128 form or k_sgemm)."""
    from llmc_amd.compression.quantization import gptq_ops
    gen = torch.Generator().manual_seed(R + K)
    X = torch.randn(2 * K, K, generator=gen)
    H = ((X.T @ X) / K).cuda()
    H += 0.01 * H.diag().mean() * torch.eye(K, device='cuda')
    U = gptq_ops.chol_inv_upper(H, check=False)
    W = (torch.randn(R, K, generator=gen) * 0.02).cuda()
    res = {}
    for name, opts in (('default', {}), ('wide4', dict(sgemm_no_wide=4)), ('k_sgemm', dict(sgemm_no_wide=1))):
        with _ffi.option(**opts), _ffi.helper_streams(False):
            res[name] = gptq_ops.gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128)
    for name in ('wide4', 'k_sgemm'):
        for a, b in zip(res['default'], res[name]):
            assert (a is None and b is None) or torch.equal(a, b), name


def test_factor_same_bits_with_far_updates_on_gemm3w():
    """K3 (gptq.py:169-176) with every eligible far update on k_gemm3w (threshold lowered so that the K = 4096 factorisation reaches it)
    against k_gemm3s: the same upper factor of the inverse, bit for bit."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    K = 4096
    gen = torch.Generator().manual_seed(7)
    X = torch.randn(2 * K, K, generator=gen)
    X[:, ::7] *= 3
    H = ((X.T @ X) / K).cuda()
    H += 0.01 * H.diag().mean() * torch.eye(K, device='cuda')
    with _ffi.option(gemm3s_min_tiles=1), _ffi.helper_streams(False):
        u_w = chol_inv_upper(H.clone())
        with _ffi.option(gemm3_no_wide=1):
            u_s = chol_inv_upper(H.clone())
    assert torch.equal(u_w, u_s)


def test_hessian_prep_vs_oracle():
    from llmc_amd.compression.quantization.gptq_ops import hessian_prep
    g = load_golden('gptq')
    p = 'asym_g128_act_dyn/'
    H = g[p + 'H']
    perm = g[p + 'perm']
    K = H.shape[0]
    W0 = g[p + 'W0']
    Hd = cu(H)
    Hout, Wout = hessian_prep(Hd, cu(W0, torch.bfloat16), torch.from_numpy(perm).cuda(), 0.01)
    np.testing.assert_array_equal(Wout.cpu().numpy(), g[p + 'Wp'])        # permuted + dead-zeroed weights
    Href = H.copy()
    dead = np.diag(Href) == 0
    assert dead.sum() == 2
    Href[dead, dead] = 1
    Href = Href[perm][:, perm]
    damp = 0.01 * np.mean(np.diag(Href).astype(np.float64))
    ho = Hout.cpu().numpy()
    off = ~np.eye(K, dtype=bool)
    np.testing.assert_array_equal(ho[off], Href[off])
    np.testing.assert_allclose(np.diag(ho), np.diag(Href) + damp, rtol=1e-6)
    assert Hd.cpu().numpy()[dead, dead].tolist() == [1.0, 1.0]           # in-place dead fix like the reference


@pytest.mark.parametrize('K', [14336, 2560, 2432, 6144, 5248])
def test_chol_inv_upper_large_property(K):
    """Uneven doubling levels (14336 = 7 * 2048): U H U^T = I in fp32 on the GPU."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    gen = torch.Generator(device='cuda').manual_seed(K)
    X = torch.randn(2 * K, K, generator=gen, device='cuda')
    H = (X.T @ X) / K + 0.05 * torch.eye(K, device='cuda')
    del X
    U = chol_inv_upper(H.clone())
    assert torch.equal(U, torch.triu(U))
    E = U @ H @ U.T
    E.diagonal().sub_(1.0)
    assert E.abs().max().item() < 1e-3


def test_owq_column_loop_and_layer_finish_match_reference_golden():
    """OWQ (gptq.py:44-56, 66-83): outlier columns last, never quantized, still fed the error; groups clipped at
    columns - n_out. Given the reference's own Hinv the loop is bit-exact; perm, final weights, merged qparams and the
    deploy-time fake quantization (fp outlier columns restored) follow."""
    import types

    from llmc_amd.compression.quantization import IntegerQuantizer
    from llmc_amd.compression.quantization.gptq import GPTQ
    from llmc_amd.compression.quantization.gptq_ops import gptq_quantize
    TDm = {'f16': torch.float16, 'bf16': torch.bfloat16, 'torch.float16': torch.float16, 'torch.bfloat16': torch.bfloat16}
    g = load_golden('gptq_owq')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        bit, sym, gs, R, K, n_out, qmin, qmax = g[p + 'meta']
        bit, sym, gs, R, K, n_out = int(bit), bool(sym), int(gs), int(R), int(K), int(n_out)
        nn_ = K - n_out
        # hessian_sorting, OWQ form (actorder is forced off): non-outlier columns in original order, then the n_out
        # largest Hessian diagonals in descending order
        perm = GPTQ.owq_permutation(cu(g[p + 'Hdiag']), n_out)
        np.testing.assert_array_equal(perm.cpu().numpy(), g[p + 'perm'], err_msg=name)
        ng = 1 if gs == 0 else K // gs
        if gs:
            init_s = cu(g[p + 'rtn_scales']).reshape(R, ng)
            init_z = cu(g[p + 'rtn_zeros']).reshape(R, ng) if g[p + 'rtn_zeros'].size else None
            Wd = cu(g[p + 'Wp'])
            tmp, losses, s, z = gptq_quantize(Wd, cu(g[p + 'U']), sym, qmin, qmax, gs, n_quant=nn_,
                                              init_scales=init_s, init_zeros=init_z)
        else:
            # per_channel OWQ: qparams of the permuted non-outlier columns (gptq.py:157-164), fp32 weights
            q32 = IntegerQuantizer(bit, sym, 'per_channel')
            _, s_pc, z_pc, _, _ = q32.get_tensor_qparams(cu(g[p + 'Wp'])[:, :nn_].contiguous())
            Wd = cu(g[p + 'Wp'])
            tmp, losses, s, z = gptq_quantize(Wd, cu(g[p + 'U']), sym, qmin, qmax, 0, scales=s_pc,
                                              zeros=None if sym else z_pc, n_quant=nn_)
        np.testing.assert_array_equal(bits(tmp.cpu().numpy()), bits(g[p + 'tmp']), err_msg=name)
        np.testing.assert_array_equal(bits(losses.cpu().numpy()), bits(g[p + 'losses']), err_msg=name)
        np.testing.assert_array_equal(bits(Wd.cpu().numpy()[:, nn_:]), bits(g[p + 'W_after'][:, nn_:]), err_msg=name)
        # finish like update_layer_with_transformed_weights (gptq.py:186-196)
        final = tmp.clone()
        final[:, nn_:] = Wd[:, nn_:]
        final = final[:, torch.argsort(perm)]
        np.testing.assert_array_equal(bits(final.cpu().numpy()), bits(g[p + 'final_w']), err_msg=name)
        if gs:
            np.testing.assert_array_equal(bits(s.reshape(-1).cpu().numpy()), bits(g[p + 'buf_scales']), err_msg=name)
            if not sym:
                np.testing.assert_array_equal(bits(z.reshape(-1).cpu().numpy()), bits(g[p + 'buf_zeros']), err_msg=name)
        # w_qdq with OWQ: fake-quant in permuted order, fp outlier columns restored, un-permuted, model dtype
        layer = torch.nn.Linear(K, R, bias=False)
        layer.weight.data = final.clone()
        sdt = torch.float32
        layer.register_buffer('buf_scales', cu(g[p + 'buf_scales']).to(sdt).reshape(-1, 1))
        if g[p + 'buf_zeros'].size:
            layer.register_buffer('buf_zeros', cu(g[p + 'buf_zeros']).to(sdt).reshape(-1, 1))
        else:
            layer.register_buffer('buf_zeros', torch.tensor(0.0))
        layer.register_buffer('buf_qmax', torch.tensor(qmax))
        layer.register_buffer('buf_qmin', torch.tensor(qmin))
        layer.register_buffer('buf_perm', perm)
        layer.register_buffer('buf_invperm', torch.argsort(perm))
        layer.register_buffer('buf_n_nonout', torch.tensor(nn_))
        layer = layer.cuda()
        gran = str(g[p + 'gran'])
        wq = IntegerQuantizer(bit, sym, gran, **({'group_size': gs} if gs else {}))
        this = types.SimpleNamespace(need_perm=True, owq=True, model_dtype=TDm[str(g[p + 'dt'])])
        fq = GPTQ.w_qdq(this, layer, wq)
        assert fq.dtype == TDm[str(g[p + 'w_qdq_dtype'])]
        np.testing.assert_array_equal(bits(fq.float().cpu().numpy()), bits(g[p + 'w_qdq']), err_msg=name)


@pytest.mark.parametrize('K', [1536, 2432, 4096, 5248])
def test_factorisation_helper_stream_changes_no_bit_and_failure_flag_survives(K):
    """llmc_chol_inv_upper with its helper stream (far updates of an outer block beside the next block's factor steps) and
    without: the same kernels in the same per-element order, so the factor must be equal bit for bit (a missing dependency
    would show as a difference or as run-to-run noise); also from a caller's side stream; and a non-positive pivot is
    reported through the flag."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    gen = torch.Generator(device='cuda').manual_seed(K)
    X = torch.randn(2 * K, K, generator=gen, device='cuda')
    H = (X.T @ X) / K
    H.diagonal().add_(0.05)
    del X
    with _ffi.helper_streams(False):
        U0 = chol_inv_upper(H.clone(), check=False).clone()
    U1 = chol_inv_upper(H.clone(), check=False).clone()
    assert torch.equal(U0, U1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        U2 = chol_inv_upper(H.clone(), check=False).clone()
        U3 = chol_inv_upper(H.clone(), check=False).clone()
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(U0, U2) and torch.equal(U0, U3)
    Hbad = H.clone()
    Hbad[K // 2, K // 2] = -1.0
    _, info = chol_inv_upper(Hbad, check=False, return_info=True)
    assert int(info.item()) != 0


@pytest.mark.parametrize('shape', [(512, 4096), (4096, 2048), (192, 5248)])
def test_pipelined_column_loop_is_bit_identical_to_the_single_stream_schedule(shape):
    """K4's far updates on the bulk stream (columns of the group after next first) against everything on one stream."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper, gptq_quantize
    R, K = shape
    gen = torch.Generator(device='cuda').manual_seed(R + K)
    X = torch.randn(2 * K, K, generator=gen, device='cuda')
    H = (X.T @ X) / K
    H.diagonal().add_(0.05)
    U = chol_inv_upper(H, check=False)
    W = torch.randn(R, K, generator=gen, device='cuda') * 0.02
    with _ffi.helper_streams(False):
        ref = gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128)
    for _ in range(2):
        out = gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128)
        for a, b in zip(ref, out):
            assert torch.equal(a, b)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = gptq_quantize(W.clone(), U, False, 0.0, 15.0, 128)
    torch.cuda.current_stream().wait_stream(side)
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(512, 4096, 3584), (256, 5248, 2944), (384, 4096, 1000)])
def test_pipelined_owq_column_loop_is_bit_identical_to_the_single_stream_schedule(shape):
    """OWQ (n_quant < K) with helper streams: the last group's per-block updates write the never-visited columns
    [n_quant, K), which earlier groups' far-far updates on the bulk stream write too (a missing dependency
    whenever last_group_start + 512 < K). Bit-identical to the single-stream schedule, repeatedly."""
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper, gptq_quantize
    R, K, nq = shape
    gen = torch.Generator(device='cuda').manual_seed(R + K + nq)
    X = torch.randn(2 * K, K, generator=gen, device='cuda')
    H = (X.T @ X) / K
    H.diagonal().add_(0.05)
    U = chol_inv_upper(H, check=False)
    W = torch.randn(R, K, generator=gen, device='cuda') * 0.02
    ng = (K + 127) // 128
    s0 = torch.ones(R, ng, device='cuda')
    z0 = torch.zeros(R, ng, device='cuda')

    def run():
        Wc = W.clone()
        out = gptq_quantize(Wc, U, False, 0.0, 15.0, 128, n_quant=nq, init_scales=s0, init_zeros=z0)
        return (Wc,) + tuple(out)       # the running weights hold the outlier columns

    with _ffi.helper_streams(False):
        ref = run()
    for _ in range(4):
        for a, b in zip(ref, run()):
            assert torch.equal(a, b)


@pytest.mark.parametrize('K', [1536, 4096, 5248])
@pytest.mark.parametrize('with_perm', [True, False])
def test_reversed_prep_and_in_place_factor_give_the_same_bits(K, with_perm):
    """llmc_hessian_prep_rev + llmc_chol_inv_upper_rev (round 5: the permuted Hessian is gathered index-reversed, the
    factorisation starts from it in place, one K^2 pass less) against llmc_hessian_prep + llmc_chol_inv_upper: the same U bit
    for bit, the same gathered weights, the same failure flag; and quantize_stacked gives the same layer either way."""
    import os
    from llmc_amd.compression.quantization import gptq_ops
    from llmc_amd.compression.quantization.gptq_pipeline import GptqConfig, quantize_stacked
    gen = torch.Generator(device='cuda').manual_seed(K + int(with_perm))
    X = torch.randn(2 * K, K, generator=gen, device='cuda') * torch.exp(0.5 * torch.randn(K, generator=gen, device='cuda'))
    H = (X.T @ X) / K
    H = torch.triu(H) + torch.triu(H, 1).T          # exactly symmetric, like every accumulated Hessian
    H[:, 7] = 0.0
    H[7, :] = 0.0                                   # a dead channel
    del X
    W = (torch.randn(320, K, generator=gen, device='cuda') * 0.02).to(torch.bfloat16)
    perm = torch.argsort(torch.diagonal(H), descending=True) if with_perm else None
    Hp, Wp = gptq_ops.hessian_prep(H.clone(), W, perm, 0.01)
    U0, i0 = gptq_ops.chol_inv_upper(Hp, check=False, return_info=True)
    U0 = U0.clone()
    Hr, Wr = gptq_ops.hessian_prep(H.clone(), W, perm, 0.01, reverse_h=True)
    assert torch.equal(Wp, Wr)
    assert torch.equal(Hr, torch.flip(gptq_ops.hessian_prep(H.clone(), None, perm, 0.01)[0], (0, 1)))
    U1, i1 = gptq_ops.chol_inv_upper_rev(Hr, check=False, return_info=True)
    assert int(i0.item()) == 0 and int(i1.item()) == 0
    assert torch.equal(U0, U1)
    cfg = GptqConfig(bit=4, symmetric=False, group_size=128, actorder=with_perm, static_groups=False)
    outs = []
    for flag in ('0', '1'):
        with _ffi.option(k3_fused_prep=int(flag)):
            r = quantize_stacked([W], H.clone(), cfg)[0]
        outs.append((r.weight.clone(), r.scales.clone(), r.zeros.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    Hbad = H.clone()
    Hbad[K // 2, K // 2] = -5.0
    Hr, _ = gptq_ops.hessian_prep(Hbad, None, perm, 0.0, reverse_h=True)
    _, info = gptq_ops.chol_inv_upper_rev(Hr, check=False, return_info=True)
    assert int(info.item()) != 0
