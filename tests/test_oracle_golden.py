"""Pin the numpy oracle to outputs of the reference itself (tests/golden/*.npz, made by
oracle/make_golden.py from /root/reference on CPU). Bit-exact for every integer/elementwise chain."""
import numpy as np

from conftest import load_golden
from oracle import quant_ref as Q


def _rows(w, gran, gs):
    return Q.reshape_rows(w, gran, gs if gs else None)


def test_quant_dynamic_cases_bit_exact():
    g = load_golden('quant')
    n = int(g['n_cases'])
    assert n >= 21
    for ci in range(n):
        p = f'c{ci}_'
        bit, sym, gs, qmin, qmax = g[p + 'meta']
        sym, gs = bool(sym), int(gs)
        dt, gran = str(g[p + 'dt']), str(g[p + 'gran'])
        w = g[p + 'w']
        w2 = _rows(w, gran, gs)
        s, z = Q.minmax_qparams(w2, dt, sym, qmin, qmax)
        tag = f'case {ci}: {dt} bit={bit} sym={sym} {gran} g={gs}'
        np.testing.assert_array_equal(s.reshape(-1).view(np.uint32), g[p + 'scales'].view(np.uint32), err_msg=tag)
        if not sym:
            np.testing.assert_array_equal(z.reshape(-1), g[p + 'zeros'], err_msg=tag)
        fq, _, _ = Q.fake_quant_dynamic(w2, dt, sym, qmin, qmax)
        np.testing.assert_array_equal(fq.reshape(w.shape).view(np.uint32), g[p + 'fake'].view(np.uint32),
                                      err_msg=tag)
        codes, rs, rz = Q.real_quant_dynamic(w2, dt, sym, qmin, qmax)
        np.testing.assert_array_equal(codes.reshape(w.shape), g[p + 'codes'], err_msg=tag)
        np.testing.assert_array_equal(rs.reshape(g[p + 'rscales'].shape), g[p + 'rscales'], err_msg=tag)
        if not sym:
            np.testing.assert_array_equal(rz.reshape(g[p + 'rzeros'].shape), g[p + 'rzeros'], err_msg=tag)


def test_quant_static_mixed_dtypes_bit_exact():
    g = load_golden('quant')
    for ci in range(int(g['n_static'])):
        p = f's{ci}_'
        bit, sym, gs, qmin, qmax = g[p + 'meta']
        wdt, sdt, zdt = [str(x) for x in g[p + 'dts']]
        zdt = None if zdt == 'none' else zdt
        w = g[p + 'w']
        w2 = w.reshape(-1, int(gs))
        s = g[p + 'scales'].reshape(-1, 1)
        z = g[p + 'zeros'].reshape(-1, 1) if zdt else None
        fq = Q.fake_quant_static(w2, wdt, s, sdt, z, zdt, qmin, qmax)
        np.testing.assert_array_equal(fq.reshape(w.shape).view(np.uint32), g[p + 'fake'].view(np.uint32),
                                      err_msg=f'static case {ci} {wdt}/{sdt}/{zdt}')
        codes, _ = Q.quant_codes(w2, wdt, s, sdt, z, zdt, qmin, qmax)
        np.testing.assert_array_equal(codes.reshape(w.shape).astype(np.int32), g[p + 'codes'])


def test_mse_range_search_bit_exact():
    """calib_algo='mse' (quant.py:145-203): the oracle reproduces the reference's searched ranges — including its
    compounding-shrink aliasing — its fp32 qparams and the fake-quantized weights on every golden row."""
    g = load_golden('mse')
    for ci, c in enumerate(g['cases']):
        dt, bit, sym, gran, gs = str(c).split('|')
        sym, gs = sym == 'True', (None if gs == 'None' else int(gs))
        p = f'c{ci}_'
        _, _, _, qmin, qmax = g[p + 'meta']
        w = g[p + 'w']
        x = _rows(w, gran, gs)
        mn, mx = Q.mse_range(x, sym, qmin, qmax)
        np.testing.assert_array_equal(mn.view(np.uint32), g[p + 'min'].view(np.uint32), err_msg=str(c))
        np.testing.assert_array_equal(mx.view(np.uint32), g[p + 'max'].view(np.uint32), err_msg=str(c))
        assert (g[p + 'min'] != g[p + 'min0']).any()          # the search moved ranges, the fixture is not trivial
        s, z = Q.qparams_from_minmax(mn, mx, Q.F32, sym, qmin, qmax)
        assert str(g[p + 'scales_dtype']) == 'torch.float32'
        np.testing.assert_array_equal(s.view(np.uint32), g[p + 'scales'].view(np.uint32), err_msg=str(c))
        if not sym:
            np.testing.assert_array_equal(z, g[p + 'zeros'], err_msg=str(c))
        fq = Q.fake_quant_static(x, dt, s[:, None], Q.F32, None if sym else z[:, None], None if sym else Q.F32, qmin, qmax)
        np.testing.assert_array_equal(fq.reshape(w.shape).view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(c))


def test_pack_vllm_bit_exact():
    g = load_golden('pack')
    for ci in range(int(g['n_vllm'])):
        packed = Q.pack_lsb(g[f'v{ci}_codes'], int(g[f'v{ci}_bit']))
        np.testing.assert_array_equal(packed, g[f'v{ci}_packed'])


def test_pack_awq_gemm_bit_exact():
    g = load_golden('pack')
    for ci in range(int(g['n_awq'])):
        qw, sc, qz = Q.pack_awq_gemm(g[f'a{ci}_w'], g[f'a{ci}_scales'], g[f'a{ci}_zeros'], int(g[f'a{ci}_g']))
        np.testing.assert_array_equal(qw, g[f'a{ci}_qweight'])
        np.testing.assert_array_equal(qz, g[f'a{ci}_qzeros'])
        np.testing.assert_array_equal(sc, g[f'a{ci}_qscales'])


def test_known_first_word_of_survey_probe():
    # nibbles (LSB first) 8,12,2,2,5,10,8,14 <=> word 0xe8a522c8 for codes+8
    codes = np.array([[0, 4, -6, -6, -3, 2, 0, 6]], dtype=np.int32)
    assert Q.pack_lsb(codes, 4).view(np.uint32)[0, 0] == 0xe8a522c8


# ---- GPTQ -------------------------------------------------------------------------------------------
from oracle import gptq_ref as G  # noqa: E402


def _gptq_cases():
    g = load_golden('gptq+more')
    return g, [str(n) for n in g['names']]


def test_sgemm_chain_model_matches_mkl_bitwise():
    g, _ = _gptq_cases()
    out = G.mm_chain(g['mm_a'], g['mm_b'])
    np.testing.assert_array_equal(out.view(np.uint32), g['mm_out'].view(np.uint32))


def test_gptq_column_loop_bit_exact_vs_reference():
    """Given the reference's own permuted weights and Hinv, the C restatement reproduces tmp, Losses and
    every group's scale/zero bit for bit."""
    g, names = _gptq_cases()
    for name in names:
        p = name + '/'
        bit, sym, gs, actorder, static_groups, R, K, qmin, qmax = g[p + 'meta']
        sym, gs, static_groups, K = bool(sym), int(gs), bool(static_groups), int(K)
        perm = g[p + 'perm']
        scales = zeros = col_group = None
        if static_groups or gs == 0:
            ng = 1 if gs == 0 else K // gs
            scales = g[p + 'buf_scales'].reshape(int(R), ng)
            zeros = g[p + 'buf_zeros'].reshape(int(R), ng) if g[p + 'buf_zeros'].size else None
            if gs:
                idx = perm if perm.size else np.arange(K)
                col_group = (idx // gs).astype(np.int32)
        r = G.weight_transform(g[p + 'Wp'], g[p + 'U'], sym, qmin, qmax, gs, static_groups, col_group,
                               scales, zeros)
        np.testing.assert_array_equal(r['tmp'].view(np.uint32), g[p + 'tmp'].view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(r['losses'].view(np.uint32), g[p + 'losses'].view(np.uint32), err_msg=name)
        if not static_groups and gs:
            np.testing.assert_array_equal(r['scales'].view(np.uint32), g[p + 'g_scales'].view(np.uint32), err_msg=name)
            if not sym:
                np.testing.assert_array_equal(r['zeros'], g[p + 'g_zeros'], err_msg=name)


def test_gptq_hessian_and_factor_within_tolerance():
    g, names = _gptq_cases()
    for name in ('asym_g128_act_dyn', 'sym_g128_act_static'):
        p = name + '/'
        K = int(g[p + 'meta'][6])
        H = np.zeros((K, K), dtype=np.float32)
        n = 0
        for x in g[p + 'x']:
            H, n = G.add_batch(H, n, x)
        ref = g[p + 'H']
        d = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-30
        # dead columns have an exactly zero row/col in both
        assert (np.abs(H - ref) / np.where(d > 0, d, 1)).max() < 1e-5
        perm = g[p + 'perm'] if g[p + 'perm'].size else None
        if perm is not None:
            np.testing.assert_array_equal(np.sort(G.hessian_sorting(ref)), np.arange(K))
        Wp, U = G.process_hessian_and_weights(g[p + 'W0'], ref, perm, 0.01)
        np.testing.assert_array_equal(Wp, g[p + 'Wp'])
        Uref = g[p + 'U']
        assert np.abs(U - Uref).max() / np.abs(Uref).max() < 2e-4, name


# ---- AWQ --------------------------------------------------------------------------------------------
from oracle import awq_ref as A  # noqa: E402


def _ulp_close(a, b, dt, max_frac_diff=0.0, max_ulps=1):
    """16-bit results of fp32 reductions: identical except where a different summation order flips the
    final rounding; then they differ by one unit in the last place."""
    a = np.asarray(a, dtype=np.float32).ravel()
    b = np.asarray(b, dtype=np.float32).ravel()
    ne = a != b
    if ne.mean() > max_frac_diff:
        return False
    if not ne.any():
        return True
    shift = 16 if dt == 'bf16' else 13
    ia = a.view(np.int32)[ne] >> shift
    ib = b.view(np.int32)[ne] >> shift
    return np.abs(ia - ib).max() <= max_ulps


def test_awq_elementwise_chain_bit_exact():
    g = load_golden('awq+more')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, nl, K, bit = [int(v) for v in g[p + 'meta']]
        dt, ver = str(g[p + 'dt']), str(g[p + 'ver'])
        ws = [g[p + f'w{i}'] for i in range(nl)]
        qmin, qmax = Q.int_range(bit, bool(sym))
        s = g[p + 'scales_r035']
        wq = np.concatenate([A.fake_quantize_weight(w, s, dt, bool(sym), qmin, qmax, gs) for w in ws], axis=0)
        np.testing.assert_array_equal(wq.view(np.uint32), g[p + 'wq_r035'].view(np.uint32), err_msg=name)
        xs = A.scaling_input(g[p + 'x'], s, dt)
        np.testing.assert_array_equal(xs.view(np.uint32), g[p + 'xs_r035'].view(np.uint32), err_msg=name)
        # get_scales from the reference's own means: powf may differ in the last fp32 bit between libms
        s2 = A.get_scales(g[p + 'x_mean'], g[p + 'w_max'], 0.5, dt, ver)
        assert _ulp_close(s2, g[p + 'scales_r050'], dt, max_frac_diff=0.02), name


def test_awq_reductions_and_search_match_reference():
    g = load_golden('awq+more')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, nl, K, bit = [int(v) for v in g[p + 'meta']]
        dt, ver = str(g[p + 'dt']), str(g[p + 'ver'])
        ws = [g[p + f'w{i}'] for i in range(nl)]
        assert _ulp_close(A.act_mean(g[p + 'x'], dt), g[p + 'x_mean'], dt, max_frac_diff=0.02), name
        assert _ulp_close(A.weight_scale(ws, dt, gs), g[p + 'w_max'], dt, max_frac_diff=0.05, max_ulps=2), name
        qmin, qmax = Q.int_range(bit, bool(sym))
        best, losses, n = A.search_scale(ws, g[p + 'x'], dt, bool(sym), qmin, qmax, gs, ver)
        ref_losses = g[p + 'losses']
        assert len(ref_losses) == 20
        # the 20 losses agree with the reference's to 1e-4 (measured: 2e-7 typical, 4.5e-5 worst: summation order of the
        # CPU GEMM behind F.linear); the argmin gaps of these goldens are 1-8 %. W8: the quantization error is so small that
        # the loss is a difference of nearly equal bf16 outputs, where one output rounding the other way shows (1.8e-4)
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-4 if bit < 8 else 5e-4, err_msg=name)
        assert n == int(np.argmin(ref_losses)), name
        # a 1-ulp difference of the token mean at the max/min channel moves the normaliser sqrt(max*min) and
        # with it every scale by one unit in the last place: same grid point, scales within 2 ulp of the dtype
        assert _ulp_close(best, g[p + 'best_scales'], dt, max_frac_diff=0.01, max_ulps=1), name
        # the search must be non-trivial on this data (interior argmin)
        assert 0 < n < 19


def test_fp8_e4m3_bit_exact_vs_torch_cast_path():
    g = load_golden('fp8')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        dt, gran = str(g[p + 'dt']), str(g[p + 'gran'])
        w = g[p + 'w']
        w2 = w.reshape(1, -1) if gran == 'per_tensor' else w
        b, s, sdt = Q.fp8_quant(w2, dt)
        np.testing.assert_array_equal(s.reshape(-1).view(np.uint32), g[p + 'scales'].view(np.uint32))
        np.testing.assert_array_equal(b.reshape(w.shape), g[p + 'bits'])
        fake = Q.fp8_fake(w2, dt).reshape(w.shape)
        np.testing.assert_array_equal(fake.view(np.uint32), g[p + 'fake'].view(np.uint32))


def test_fp8_qtorch_semantics_e4m3_and_e5m2_bit_exact():
    """fp8_qtorch.npz: llmc's FloatQuantizer with float_quantize bound to the restated qtorch algorithm. The oracle's
    restatement of everything AROUND that call (scales from finfo.max, division and +0 in the tensor dtype, the exact cast of
    the quantized values, the fp32 dequantisation product) must reproduce the reference class bit for bit, for both formats."""
    g = load_golden('fp8_qtorch')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        dt, gran, bit = str(g[p + 'dt']), str(g[p + 'gran']), str(g[p + 'bit'])
        w = g[p + 'w']
        w2 = w.reshape(1, -1) if gran == 'per_tensor' else w
        b, s, sdt = Q.fp8_quant(w2, dt, bit, 'qtorch')
        np.testing.assert_array_equal(s.reshape(-1).view(np.uint32), g[p + 'scales'].view(np.uint32))
        np.testing.assert_array_equal(b.reshape(w.shape), g[p + 'bits'], err_msg=str(ci))
        fake = Q.fp8_fake(w2, dt, bit, 'qtorch').reshape(w.shape)
        np.testing.assert_array_equal(fake.view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(ci))
    g = load_golden('fp8_block_qtorch')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        bits, scales, fake = Q.fp8_per_block(g[p + 'w'], str(g[p + 'dt']), int(g[p + 'block']), 'qtorch')
        np.testing.assert_array_equal(scales.view(np.uint32), g[p + 'scales'].view(np.uint32), err_msg=str(ci))
        np.testing.assert_array_equal(bits, g[p + 'bits'], err_msg=str(ci))
        np.testing.assert_array_equal(fake.view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(ci))


def _fp8_boundary_patterns(man, lo_exp, hi_exp):
    """every fp32 whose leading bits select a grid point, a midpoint or a quarter point of an (e, man) grid, +-2 ulp, both signs"""
    vals = []
    for e in range(lo_exp, hi_exp + 1):
        for m in range(0, 1 << (man + 2)):
            base = ((e + 127) << 23) | (m << (23 - man - 2))
            vals.extend(base + d for d in (-2, -1, 0, 1, 2))
    v = np.array(vals, dtype=np.uint32)
    x = np.concatenate([v, v | np.uint32(0x80000000)]).view(np.float32)
    return x[np.isfinite(x)]


def test_qtorch_vs_torch_cast_where_they_differ():
    """The restated qtorch.float_quantize(x, 4, 3) / (x, 5, 2) against torch's float8 casts over every boundary pattern of
    the grids: they agree everywhere EXCEPT (a) exact ties (away from zero vs to even), (b) e4m3 from
    |x| >= 248 on — qtorch's IEEE-style e4m3 ends at 240 and saturates, the OCP e4m3fn of llmc's qmax goes on to 448 and
    turns NaN above 464 —, (c) e5m2 above 61440 (57344 vs inf), (d) inputs one or two ulp below a midpoint of the SUBNORMAL
    grid, where qtorch's `x + 2^min_exp` addition rounds first (a double rounding). Also pins the oracle's own bit-level
    casts to torch's."""
    import torch
    x = np.concatenate([_fp8_boundary_patterns(3, -12, 9), np.array([0.0, -0.0], np.float32)])
    qt = Q.qtorch_float_quantize(x, 4, 3)
    tc = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    mine = Q.e4m3fn_bits_to_f32(Q.f32_to_e4m3fn_bits(x))
    assert np.array_equal(np.nan_to_num(mine, nan=-1.0), np.nan_to_num(tc, nan=-1.0))       # oracle cast == torch cast
    ax = np.abs(x)
    same = (qt == tc) | (np.isnan(qt) & np.isnan(tc))
    tie = (ax.view(np.uint32) & np.uint32((1 << 20) - 1)) == np.uint32(1 << 19)
    normal = (ax >= 2.0 ** -6) & (ax < 248.0)
    assert same[normal & ~tie].all()                       # off the ties the two roundings agree on the common range
    assert (~same[normal & tie]).sum() > 0 and (np.abs(qt[normal & tie]) >= np.abs(tc[normal & tie])).all()   # away from zero
    assert (~same[ax >= 248.0]).all() and (np.abs(qt[ax >= 248.0]) == 240.0).all()           # saturation at 240
    sub = ax < 2.0 ** -6
    assert (np.abs(qt[sub] - tc[sub]) <= 2.0 ** -9 + 1e-12).all() and same[sub].mean() > 0.97  # same grid, a neighbour at most
    assert np.array_equal(Q.qtorch_float_quantize(np.float32([448.0, -448.0, 240.0, 247.9, 248.0]), 4, 3),
                          np.float32([240.0, -240.0, 240.0, 240.0, 240.0]))
    x5 = _fp8_boundary_patterns(2, -18, 16)
    q5 = Q.qtorch_float_quantize(x5, 5, 2)
    t5 = torch.from_numpy(x5).to(torch.float8_e5m2).float().numpy()
    m5 = Q.e5m2_bits_to_f32(Q.f32_to_e5m2_bits(x5))
    assert np.array_equal(np.nan_to_num(m5, nan=-1.0, posinf=1e30, neginf=-1e30), np.nan_to_num(t5, nan=-1.0, posinf=1e30, neginf=-1e30))
    a5 = np.abs(x5)
    same5 = (q5 == t5)
    tie5 = (a5.view(np.uint32) & np.uint32((1 << 21) - 1)) == np.uint32(1 << 20)
    n5 = (a5 >= 2.0 ** -14) & (a5 < 61440.0)
    assert same5[n5 & ~tie5].all() and (np.abs(q5[a5 >= 61440.0]) == 57344.0).all()


def test_auto_clip_matches_reference():
    """auto_clip_layer restated with the reference's roundings chooses the reference's clip level for EVERY (row, group)
    of the goldens: one batch (clip.npz) and the list form (clip_mb.npz, error averaged over the batches)."""
    g = load_golden('clip+more')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, clip_sym, nst, bit = [int(v) for v in g[p + 'meta']]
        dt = str(g[p + 'dt'])
        qmin, qmax = Q.int_range(bit, bool(sym))
        mx, mn = A.auto_clip_layer(g[p + 'w'], g[p + 'x'], dt, bool(sym), qmin, qmax, gs, bool(clip_sym),
                                   n_sample_token=nst)
        ref_mx, ref_mn = g[p + 'best_max'], g[p + 'best_min']
        np.testing.assert_array_equal(mx.reshape(ref_mx.shape), ref_mx, err_msg=name)
        np.testing.assert_array_equal(mn.reshape(ref_mn.shape), ref_mn, err_msg=name)
        # clipped values are always one of the 10 candidate levels of the group's original max
        assert (ref_mx > 0).all()
    g = load_golden('clip_mb')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, clip_sym, nst, nb = [int(v) for v in g[p + 'meta']]
        dt = str(g[p + 'dt'])
        qmin, qmax = Q.int_range(4, bool(sym))
        xs = [g[p + f'x{i}'] for i in range(nb)]
        mx, mn = A.auto_clip_layer(g[p + 'w'], xs, dt, bool(sym), qmin, qmax, gs, bool(clip_sym), n_sample_token=nst)
        np.testing.assert_array_equal(mx.reshape(g[p + 'best_max'].shape), g[p + 'best_max'], err_msg=name)
        np.testing.assert_array_equal(mn.reshape(g[p + 'best_min'].shape), g[p + 'best_min'], err_msg=name)


def bits16_to_f32(bits, dt):
    """uint16 patterns of a 16-bit dtype (how clip_wide.npz stores the candidates) -> fp32 values."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16))
    return t.view(torch.bfloat16 if dt == 'bf16' else torch.float16).float().numpy()


def sampled_tokens(x, nst):
    """auto_clip.py:133-147 without a padding mask: flatten, every step-th token."""
    x2 = x.reshape(-1, x.shape[-1])
    return x2[0::max(1, x2.shape[0] // nst)]


def test_auto_clip_from_candidates_matches_reference_for_wide_groups_act_quant_fp8_and_v2():
    """clip_wide.npz: per_channel / per_tensor ranges (one group = the row: ATen's cascaded inner sum, K = 1152 flushes a
    level, K = 488 has tail vectors and trailing elements), quantized activations, FP8 quantizers, clip_version v2. From the
    candidates the reference formed, the restated error table picks the reference's level for EVERY (row, group)."""
    g = load_golden('clip_wide')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        R, K, gs, clip_sym, nst = [int(v) for v in g[p + 'meta']]
        dt = str(g[p + 'dt'])
        x = sampled_tokens(g[p + 'x'], nst)
        xq = bits16_to_f32(g[p + 'qx_bits'], dt)
        assert xq.shape == x.shape, name
        if g[p + 'acfg'].size == 0:
            np.testing.assert_array_equal(xq, x, err_msg=name)            # w_only: fake_quantize_input is the identity
        cands = bits16_to_f32(g[p + 'cands_bits'], dt)
        errs = A.clip_errs_from_candidates(g[p + 'w'], cands, x, xq, dt, gs)
        mx, mn = A.clip_argmin_levels(errs, g[p + 'w'], gs, dt, bool(clip_sym))
        np.testing.assert_array_equal(mx.reshape(g[p + 'best_max'].shape), g[p + 'best_max'], err_msg=name)
        np.testing.assert_array_equal(mn.reshape(g[p + 'best_min'].shape), g[p + 'best_min'], err_msg=name)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_awq_with_activation_quantization_chain_point_bit_exact():
    """awq_wa.npz: the scaled + fake-quantized weight and the scaled + fake-quantized input of one grid point, as the
    reference formed them for W8A8 / W4A8 (per-sample per_tensor activations) / FP8 e4m3 per_tensor (awq_fp8_static.yml) /
    FP8 e5m2 / per_tensor INT8 weights — restated with the oracle's quantizers (oracle/awq_ref.py:wa_chain_point)."""
    g = load_golden('awq_wa')

    def cfg(a):
        a = [str(v) for v in a]
        out = [a[0], a[1] if a[0] == 'float' else int(a[1]), a[2] == 'True', a[3]]
        if len(a) > 4:
            out.append(int(a[4]))
        return tuple(out)
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        K, R, nb, awq_bs = [int(v) for v in g[p + 'meta']]
        dt = str(g[p + 'dt'])
        wq, xq = A.wa_chain_point(g[p + 'w_gate_proj'], g[p + 'x0'], g[p + 'scales_r040'], dt, cfg(g[p + 'wcfg']), cfg(g[p + 'acfg']),
                                  per_sample=bool(awq_bs) and awq_bs != g[p + 'x0'].shape[0])
        np.testing.assert_array_equal(bits(wq), bits(g[p + 'wq_r040']), err_msg=name)
        np.testing.assert_array_equal(bits(xq), bits(g[p + 'xq_r040']), err_msg=name)


def test_auto_clip_general_restatement_builds_the_candidates_itself():
    """clip_wide.npz again, this time with the candidates formed by the oracle's own quantizers (v1: fake-quant of the clamped weights
    per output-channel batch; v2: learnable range from logit / sigmoid of the level ratios): the reference's level for every row."""
    g = load_golden('clip_wide')

    def cfg(a):
        a = [str(v) for v in a]
        if not a:
            return None
        out = [a[0], a[1] if a[0] == 'float' else int(a[1]), a[2] == 'True', a[3]]
        if len(a) > 4:
            out.append(int(a[4]))
        return tuple(out)
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        R, K, gs, clip_sym, nst = [int(v) for v in g[p + 'meta']]
        dt, ver = str(g[p + 'dt']), str(g[p + 'ver'])
        mx, mn = A.auto_clip_layer_general(g[p + 'w'], g[p + 'x'], dt, cfg(g[p + 'wcfg']), cfg(g[p + 'acfg']), ver, bool(clip_sym),
                                           n_sample_token=nst)
        eq_mx = (mx.reshape(g[p + 'best_max'].shape) == g[p + 'best_max']).mean()
        eq_mn = (mn.reshape(g[p + 'best_min'].shape) == g[p + 'best_min']).mean()
        assert eq_mx == 1.0 and eq_mn == 1.0, (name, eq_mx, eq_mn)


def test_awq_with_activation_quantization_search_restated_for_the_single_layer_cases():
    """awq_wa.npz, the two cases whose inspected module is the layer itself (FP8 e4m3 per_tensor = awq_fp8_static.yml's arithmetic,
    INT8 per_tensor weights with asymmetric per_token activations): the restated search (oracle/awq_ref.py:search_scale_wa) follows
    the reference's loss curve within 1e-3 (the GEMMs sum in another order) and returns its scales."""
    g = load_golden('awq_wa')

    def cfg(a):
        a = [str(v) for v in a]
        out = [a[0], a[1] if a[0] == 'float' else int(a[1]), a[2] == 'True', a[3]]
        if len(a) > 4:
            out.append(int(a[4]))
        return tuple(out)
    n_cases = 0
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        if str(g[p + 'inspect']) != 'linear':
            continue
        K, R, nb, awq_bs = [int(v) for v in g[p + 'meta']]
        dt = str(g[p + 'dt'])
        best, losses = A.search_scale_wa(g[p + 'w_gate_proj'], [g[p + f'x{i}'] for i in range(nb)], dt, cfg(g[p + 'wcfg']), cfg(g[p + 'acfg']))
        ref = g[p + 'losses']
        assert losses.shape == ref.shape
        np.testing.assert_allclose(losses, ref, rtol=1e-3, err_msg=name)
        assert int(np.argmin(losses)) == int(np.argmin(ref)), name
        u = np.abs((bits(best) >> 16).astype(np.int64) - (bits(g[p + 'best_scales']) >> 16).astype(np.int64))
        assert u.max() <= 2, (name, u.max())
        n_cases += 1
    assert n_cases == 2


def test_fp8_per_group_bit_exact():
    """fp8_group_qtorch.npz: FloatQuantizer per_group (activations of rtn_w_a_block.yml, FP8 in groups of 128) through the oracle."""
    g = load_golden('fp8_group_qtorch')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        dt, bit, gs = str(g[p + 'dt']), str(g[p + 'bit']), int(g[p + 'gs'])
        for src, ref in (('x', 'fake_x'), ('w', 'fake_w')):
            x = g[p + src]
            got = Q.fp8_fake(x.reshape(-1, gs), dt, bit, 'qtorch').reshape(x.shape)
            np.testing.assert_array_equal(bits(got), bits(g[p + ref]), err_msg=f'{ci} {src}')


def test_per_tensor_asymmetric_bit_exact():
    """quant_pt.npz: fp32 0-dim qparams, op results in the tensor dtype."""
    g = load_golden('quant_pt')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        bit, qmin, qmax = g[p + 'meta']
        dt = str(g[p + 'dt'])
        fake, codes, s, z = Q.per_tensor_asym_fake_and_codes(g[p + 'w'], dt, qmin, qmax)
        np.testing.assert_array_equal(np.array([s]).view(np.uint32), g[p + 'scales'].view(np.uint32), err_msg=str(ci))
        np.testing.assert_array_equal(np.array([z]), g[p + 'zeros'], err_msg=str(ci))
        np.testing.assert_array_equal(fake.view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(ci))
        np.testing.assert_array_equal(codes, g[p + 'codes'], err_msg=str(ci))


def test_fp8_per_block_bit_exact():
    g = load_golden('fp8_block')
    for ci in range(int(g['n'])):
        p = f'c{ci}_'
        bits, scales, fake = Q.fp8_per_block(g[p + 'w'], str(g[p + 'dt']), int(g[p + 'block']))
        np.testing.assert_array_equal(scales.view(np.uint32), g[p + 'scales'].view(np.uint32), err_msg=str(ci))
        np.testing.assert_array_equal(bits, g[p + 'bits'], err_msg=str(ci))
        np.testing.assert_array_equal(fake.view(np.uint32), g[p + 'fake'].view(np.uint32), err_msg=str(ci))
        np.testing.assert_array_equal(bits, g[p + 'cast_bits'])
        back = Q.rnd((Q.e4m3fn_bits_to_f32(bits) * np.repeat(np.repeat(scales, int(g[p + 'block']), 0), int(g[p + 'block']), 1)
                      [:bits.shape[0], :bits.shape[1]]).astype(np.float32), 'bf16')
        np.testing.assert_array_equal(back.view(np.uint32), g[p + 'cast_back'].view(np.uint32), err_msg=str(ci))


def test_gptq_owq_loop_bit_exact():
    """gptq_owq.npz: OWQ permutation and the column loop restricted to the non-outlier columns, given the reference's Hinv."""
    from oracle import gptq_ref as G
    g = load_golden('gptq_owq')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        bit, sym, gs, R, K, n_out, qmin, qmax = g[p + 'meta']
        sym, gs, R, K, n_out = bool(sym), int(gs), int(R), int(K), int(n_out)
        np.testing.assert_array_equal(G.owq_permutation(g[p + 'Hdiag'], n_out), g[p + 'perm'], err_msg=name)
        nn_ = K - n_out
        if gs:
            ng = K // gs
            r = G.weight_transform(g[p + 'Wp'], g[p + 'U'], sym, qmin, qmax, gs, n_quant=nn_,
                                   init_scales=g[p + 'rtn_scales'].reshape(R, ng),
                                   init_zeros=g[p + 'rtn_zeros'].reshape(R, ng) if g[p + 'rtn_zeros'].size else None)
        else:
            s, z = Q.minmax_qparams(g[p + 'Wp'][:, :nn_], 'f32', sym, qmin, qmax)
            r = G.weight_transform(g[p + 'Wp'], g[p + 'U'], sym, qmin, qmax, 0, scales=s, zeros=None if sym else z,
                                   n_quant=nn_)
        np.testing.assert_array_equal(r['tmp'].view(np.uint32), g[p + 'tmp'].view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(r['losses'].view(np.uint32), g[p + 'losses'].view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(r['W'][:, nn_:].view(np.uint32), g[p + 'W_after'][:, nn_:].view(np.uint32), err_msg=name)
        if gs:
            np.testing.assert_array_equal(r['scales'].reshape(-1).view(np.uint32), g[p + 'buf_scales'].view(np.uint32), err_msg=name)


def test_spqr_oracle_bit_exact_vs_reference():
    """oracle/spqr_ref.py + csrc/spqr_canon.c against the reference's SpQR.layer_transform / w_qdq (spqr.py:116-380):
    prep and threshold by tolerance (LAPACK / reduction order), the column loop — leave-one-out detection, second-level
    qparams, mask, tmp, losses — and the deploy-time fake quantization bit for bit."""
    from oracle import spqr_ref as S
    g = load_golden('spqr')
    names = sorted({k.split('/')[0] for k in g.files})
    assert len(names) == 4
    for n in names:
        p = n + '/'
        bit, gs, act, R, K, simp = [int(v) for v in g[p + 'cfg']]
        dt = 'bf16' if n == 'g32_noact_thr01' else 'f16'
        Wp, U, perm = S.process_hessian_and_weights(g[p + 'W0'], g[p + 'H'], bool(act), float(g[p + 'percdamp']))
        np.testing.assert_array_equal(Wp, g[p + 'Wp'], err_msg=n)
        if act:
            np.testing.assert_array_equal(perm, g[p + 'perm'], err_msg=n)
        assert np.abs(U - g[p + 'U']).max() <= 2e-6 * np.abs(g[p + 'U']).max(), n
        thr = S.outlier_threshold(g[p + 'Wp'], g[p + 'U'], float(g[p + 'rel_threshold']))
        ref_thr = float(g[p + 'threshold'])
        assert (np.isinf(thr) and np.isinf(ref_thr)) or abs(thr - ref_thr) <= 1e-5 * ref_thr, n
        o = S.weight_transform(g[p + 'Wp'], g[p + 'U'], bit, gs, ref_thr, bool(simp))
        np.testing.assert_array_equal(o['mask'], g[p + 'mask'], err_msg=n)
        np.testing.assert_array_equal(o['tmp'], g[p + 'tmp'], err_msg=n)
        np.testing.assert_array_equal(o['losses'], g[p + 'losses'], err_msg=n)
        np.testing.assert_array_equal(o['scales'].reshape(-1, 1), g[p + 'buf_scales'], err_msg=n)
        np.testing.assert_array_equal(o['zeros'].reshape(-1, 1), g[p + 'buf_zeros'], err_msg=n)
        wq = S.w_qdq(g[p + 'weight'], g[p + 'buf_mask'], g[p + 'buf_scales'], g[p + 'buf_zeros'], bit, gs, dt,
                     g[p + 'perm'] if act else None)
        np.testing.assert_array_equal(wq, g[p + 'w_qdq'], err_msg=n)
    # the detection branch is exercised: it changes the qparams of some groups
    p = 'g16_act_thr02/'
    a = S.weight_transform(g[p + 'Wp'], g[p + 'U'], 4, 16, float(g[p + 'threshold']), False)
    b = S.weight_transform(g[p + 'Wp'], g[p + 'U'], 4, 16, float(g[p + 'threshold']), True)
    assert (a['scales'] != b['scales']).sum() > 0


def test_static_hist_oracle_and_host_search_bit_exact_vs_reference():
    """calib_algo static_hist (quant.py:264-512): the oracle (oracle/hist_ref.py) reproduces the reference's merged
    histograms bin for bin (incl. the re-binning when the range grows) and its searched range exactly; the product's host
    search (llmc_amd/.../hist_range.py, whose data pass is llmc_histc on the GPU) gives the same range from the same
    histogram."""
    from llmc_amd.compression.quantization.hist_range import HistRange
    from oracle import hist_ref as Hs
    g = load_golden('hist')
    names = sorted({k.split('/')[0] for k in g.files})
    assert len(names) == 4
    for n in names:
        p = n + '/'
        a, b, hist, mn, mx, _, _ = Hs.static_hist_range(list(g[p + 'x']), 'bf16')
        np.testing.assert_array_equal(hist, g[p + 'hist'], err_msg=n)
        assert mn == g[p + 'min'] and mx == g[p + 'max'], n
        assert a == g[p + 'new_min'] and b == g[p + 'new_max'], (n, a, b)
        h = HistRange(2048, 16, 256)
        h.hist, h.lo, h.hi = g[p + 'hist'].copy(), np.float32(g[p + 'min']), np.float32(g[p + 'max'])
        lo, hi = h.range()
        assert lo == g[p + 'new_min'] and hi == g[p + 'new_max'], (n, lo, hi)
    # the re-binning of the product against the oracle's on the case whose range grows with every sample
    p = 'growing/'
    xs = list(g[p + 'x'])
    h = HistRange(2048, 16, 256)
    h.hist, h.lo, h.hi = Hs.histc(xs[0], 2048, xs[0].min(), xs[0].max()), np.float32(xs[0].min()), np.float32(xs[0].max())
    lo, hi = min(h.lo, np.float32(xs[1].min())), max(h.hi, np.float32(xs[1].max()))
    np.testing.assert_array_equal(h._rebin(lo, hi), Hs.upscale_histogram(h.hist, h.lo, h.hi, lo, hi))


def _from16(bits, dt):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16 if dt == 'bf16' else torch.float16)
    return t.float().numpy()


def test_fp8_block_restatement_pinned_to_the_reference_triton_kernels():
    """tests/golden/fp8_triton.npz: outputs of the reference's own Triton kernels (kernel.py:7-242: act_quant,
    weight_cast_to_fp8, fp8_gemm) run unmodified on an MI355X (tools/fp8_triton_golden.py). The restatement in
    oracle/quant_ref.py reproduces the codes and scales bit for bit and the GEMM within its summation order."""
    g = load_golden('fp8_triton')
    for dt in ('bf16', 'f16'):
        bits, s = Q.act_quant_ref(_from16(g[f'aq_{dt}_x16'], dt), 128)
        np.testing.assert_array_equal(s.view(np.uint32), g[f'aq_{dt}_scales'].view(np.uint32))
        np.testing.assert_array_equal(bits, g[f'aq_{dt}_bits'])
    for i in range(int(g['n_gemm'])):
        p = f'g{i}_'
        bits, s = Q.act_quant_ref(_from16(g[p + 'x16'], 'bf16'), 128)
        np.testing.assert_array_equal(bits, g[p + 'a_bits'])
        np.testing.assert_array_equal(s.view(np.uint32), g[p + 'a_s'].view(np.uint32))
        wb, ws, _ = Q.fp8_per_block(_from16(g[p + 'w16'], 'bf16'), 'bf16', 128)
        np.testing.assert_array_equal(wb, g[p + 'w_bits'])
        np.testing.assert_array_equal(ws.view(np.uint32), g[p + 'w_s'].view(np.uint32))
        ref = Q.fp8_block_gemm_ref(g[p + 'a_bits'], g[p + 'a_s'], g[p + 'w_bits'], g[p + 'w_s'])
        c = _from16(g[p + 'c_bf16_16'], 'bf16')
        err = np.abs(Q.rnd(ref, 'bf16') - c)
        assert (err <= 2.0 ** -7 * np.abs(ref) + 1e-4 * np.abs(ref).max()).all()      # one bf16 rounding apart at most


def test_aten_summation_orders_restated_exactly():
    """oracle/aten_sum.py against torch itself (AVX-512 builds, the capability the goldens were produced with): the
    contiguous 16-bit inner sum bit for bit in fp32-before-rounding terms (checked through the rounded result on 2e5
    rows), and the serial outer sum in fp32 bit for bit."""
    import torch
    from oracle import aten_sum as AS
    if torch.backends.cpu.get_cpu_capability() not in ('AVX512', 'AVX2'):
        pytest.skip('ATen vector width differs on this host')     # sum_stub has no AVX-512 variant: AVX2 code on both
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        for n in (128, 64, 96, 32, 16):
            x = (torch.randn(50000, n) * torch.exp(torch.randn(n))).to(dt)
            mine = torch.from_numpy(AS.inner_sum_16bit(x.float().numpy())).to(dt)
            assert torch.equal(x.sum(-1), mine), (dt, n)
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)                 # the serial iterator: what small inputs take at any thread count
    try:
        for (oc, tok, ng) in [(64, 32, 2), (64, 32, 4), (64, 100, 7), (64, 37, 16), (64, 37, 80), (32, 512, 32), (16, 300, 112)]:
            t = (torch.randn(oc, tok, ng) ** 2 * torch.exp(torch.randn(ng))).to(torch.bfloat16).float()
            assert torch.equal(t.sum(dim=1), torch.from_numpy(AS.outer_sum_fp32(t.numpy()))), (oc, tok, ng)
    finally:
        torch.set_num_threads(nthr)


def test_awq_search_with_two_near_equal_minima_picks_the_reference_grid_point():
    """tests/golden/awq_flat.npz: the reference's second-best loss is 3.7e-4 above its best. The restatement's losses agree
    to 1e-5, so the same grid point wins — nothing here relies on a percent-level tolerance."""
    g = load_golden('awq_flat')
    name = str(g['names'][0])
    p = name + '/'
    sym, gs, nl, K = [int(v) for v in g[p + 'meta']]
    dt, ver = str(g[p + 'dt']), str(g[p + 'ver'])
    ws = [g[p + f'w{i}'] for i in range(nl)]
    qmin, qmax = Q.int_range(4, bool(sym))
    best, losses, n = A.search_scale(ws, g[p + 'x'], dt, bool(sym), qmin, qmax, gs, ver)
    ref = g[p + 'losses']
    srt = np.sort(ref)
    gap = (srt[1] - srt[0]) / srt[0]
    assert 1e-4 < gap < 2e-3
    assert np.abs(losses - ref).max() / ref.min() < gap / 10
    assert n == int(np.argmin(ref))
    assert _ulp_close(best, g[p + 'best_scales'], dt, max_frac_diff=0.01, max_ulps=1)


def test_awq_fp8_checkpoint_branches_bit_exact_vs_reference():
    """oracle/awq_ref.py:fp8ckpt_* against the reference's own class code on block-wise FP8 modules (awq_fp8ckpt.npz:
    get_weight_scale, fake_quantize_weight, w_qdq, scale_ln_fcs, scale_fc_fc with the non-Triton casts)."""
    from oracle import awq_ref as A
    g = load_golden('awq_fp8ckpt')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        bit, sym, gs, bsz, K = [int(v) for v in g[p + 'meta']]
        qmin, qmax = Q.int_range(bit, bool(sym))
        layers = [(g[p + f'w8_{i}'], g[p + f's8_{i}']) for i in range(2)]
        np.testing.assert_array_equal(A.fp8ckpt_weight_scale(layers, bsz, gs).view(np.uint32), g[p + 'w_max'].view(np.uint32), err_msg=name)
        cols = g[p + 'scales']
        for i, (b8, s8) in enumerate(layers):
            fb, fs = A.fp8ckpt_fake_quantize_weight(b8, s8, cols, bsz, bool(sym), qmin, qmax, gs)
            np.testing.assert_array_equal(fs.view(np.uint32), g[p + f'fq_s8_{i}'].view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(fb, g[p + f'fq_w8_{i}'], err_msg=name)
            qb, qs = A.fp8ckpt_w_qdq(b8, s8, bsz, bool(sym), qmin, qmax, gs)
            np.testing.assert_array_equal(qs.view(np.uint32), g[p + f'qdq_s8_{i}'].view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(qb, g[p + f'qdq_w8_{i}'], err_msg=name)
            lb, ls = A.fp8ckpt_mul_cols(b8, s8, cols, bsz)
            np.testing.assert_array_equal(ls.view(np.uint32), g[p + f'ln_s8_{i}'].view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(lb, g[p + f'ln_w8_{i}'], err_msg=name)
        np.testing.assert_array_equal(Q.rnd(g[p + 'ln_w'] / cols, 'bf16').view(np.uint32), g[p + 'ln_w_after'].view(np.uint32))
        b1, s1 = A.fp8ckpt_div_rows(g[p + 'fc1_w8'], g[p + 'fc1_s8'], cols, bsz)
        np.testing.assert_array_equal(b1, g[p + 'fc1_w8_after'], err_msg=name)
        np.testing.assert_array_equal(s1.view(np.uint32), g[p + 'fc1_s8_after'].view(np.uint32), err_msg=name)
        b2, s2 = A.fp8ckpt_mul_cols(g[p + 'fc2_w8'], g[p + 'fc2_s8'], cols, bsz)
        np.testing.assert_array_equal(b2, g[p + 'fc2_w8_after'], err_msg=name)
        np.testing.assert_array_equal(s2.view(np.uint32), g[p + 'fc2_s8_after'].view(np.uint32), err_msg=name)
