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
