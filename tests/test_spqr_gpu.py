"""SpQR on MI355X: llmc_spqr_quantize against the oracle (bit-exact) and the reference goldens, the factor / threshold
pipeline, the round_zp=False static quantizer, and the SpQR class end to end against the reference's class."""
import copy
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spqr_ref as S

pytestmark = pytest.mark.gpu

CASES = ['g16_act_thr02', 'g32_noact_thr01', 'g16_act_inf', 'g64_act_simplified']


def _cfg(g, n):
    from llmc_amd.compression.quantization.spqr import SpqrConfig
    p = n + '/'
    bit, gs, act, R, K, simp = [int(v) for v in g[p + 'cfg']]
    return SpqrConfig(bit=bit, group_size=gs, actorder=bool(act), percdamp=float(g[p + 'percdamp']),
                      relative_threshold=float(g[p + 'rel_threshold']), simplified_outliers=bool(simp))


@pytest.mark.parametrize('name', CASES)
def test_column_loop_bit_exact_vs_reference_golden(name):
    from llmc_amd.compression.quantization.spqr import spqr_quantize
    g = load_golden('spqr')
    p = name + '/'
    cfg = _cfg(g, name)
    W = torch.from_numpy(g[p + 'Wp'].copy()).cuda()
    U = torch.from_numpy(g[p + 'U'].copy()).cuda()
    tmp, losses, mask, s, z = spqr_quantize(W, U, cfg, float(g[p + 'threshold']))
    np.testing.assert_array_equal(mask.cpu().numpy(), g[p + 'mask'])
    np.testing.assert_array_equal(tmp.cpu().numpy(), g[p + 'tmp'])
    np.testing.assert_array_equal(losses.cpu().numpy(), g[p + 'losses'])
    np.testing.assert_array_equal(s.cpu().numpy().reshape(-1, 1), g[p + 'buf_scales'])
    np.testing.assert_array_equal(z.cpu().numpy().reshape(-1, 1), g[p + 'buf_zeros'])


@pytest.mark.parametrize('gs,R,K,simp,thr', [(16, 200, 512, False, 0.05), (128, 70, 384, False, 0.02),
                                             (32, 33, 256, True, 0.1), (64, 48, 256, False, math.inf),
                                             (16, 1000, 1536, False, 0.1)])
def test_column_loop_bit_exact_vs_oracle_random(gs, R, K, simp, thr):
    """Sizes with ragged row counts, several blocks and all group sizes; many detected outliers (small thresholds)."""
    from llmc_amd.compression.quantization.spqr import SpqrConfig, spqr_quantize
    rs = np.random.RandomState(gs + R)
    W = (rs.randn(R, K) * 0.02).astype(np.float32)
    W[:, rs.randint(0, K, 6)] *= 15
    W[rs.randint(0, R, 40), rs.randint(0, K, 40)] *= 25           # isolated outliers inside groups
    W[3, :gs] = 0.013                                              # a constant group
    X = (rs.randn(4 * K, K) * np.exp(0.5 * rs.randn(K))).astype(np.float32)
    H = (X.T @ X / 8).astype(np.float32)
    Wp, U, _ = S.process_hessian_and_weights(W, H, True, 1.0)
    t = S.outlier_threshold(Wp, U, thr)
    o = S.weight_transform(Wp, U, 4, gs, t, simp)
    cfg = SpqrConfig(bit=4, group_size=gs, relative_threshold=thr, simplified_outliers=simp)
    tmp, losses, mask, s, z = spqr_quantize(torch.from_numpy(Wp.copy()).cuda(), torch.from_numpy(U).cuda(), cfg, t)
    if not math.isinf(thr):
        assert o['mask'].sum() > 0
    np.testing.assert_array_equal(mask.cpu().numpy(), o['mask'])
    np.testing.assert_array_equal(s.cpu().numpy(), o['scales'])
    np.testing.assert_array_equal(z.cpu().numpy(), o['zeros'])
    np.testing.assert_array_equal(tmp.cpu().numpy(), o['tmp'])
    np.testing.assert_array_equal(losses.cpu().numpy(), o['losses'])


@pytest.mark.parametrize('name', CASES)
def test_layer_pipeline_and_deploy_vs_reference_golden(name):
    """H, W0 -> permutation, damping, factor, threshold, loop, un-permutation (quantize_stacked) and w_qdq's quantizer.
    The factor differs from LAPACK's in the last bits, so the loop's decisions may flip where they are ties: bulk
    agreement; the deploy-time quantizer on the golden buffers is bit-exact."""
    from llmc_amd.compression.quantization.quant import IntegerQuantizer
    from llmc_amd.compression.quantization.spqr import quantize_stacked
    g = load_golden('spqr')
    p = name + '/'
    cfg = _cfg(g, name)
    dt = torch.bfloat16 if name == 'g32_noact_thr01' else torch.float16
    H = torch.from_numpy(g[p + 'H'].copy()).cuda()
    W0 = torch.from_numpy(g[p + 'W0']).to(dt).cuda()
    r = quantize_stacked([W0], H, cfg)[0]
    assert int(r.info.item()) == 0
    if cfg.actorder:
        ref_perm = g[p + 'perm']
        d = np.diag(g[p + 'H'])
        np.testing.assert_array_equal(d[r.perm.cpu().numpy()], d[ref_perm])      # ties (dead columns) may swap
    if not math.isinf(cfg.relative_threshold):
        assert abs(r.threshold - float(g[p + 'threshold'])) <= 1e-3 * float(g[p + 'threshold'])
    ref_w, got = g[p + 'weight'], r.weight.cpu().numpy()
    scale = np.abs(ref_w).max()
    assert np.mean(np.abs(got - ref_w) <= 1e-3 * scale) > 0.97
    assert np.mean(r.mask.cpu().numpy() == g[p + 'buf_mask'].astype(bool)) > 0.995
    s_ref = g[p + 'buf_scales'].reshape(-1)
    assert np.mean(np.abs(r.scales.cpu().numpy().reshape(-1) - s_ref) <= 1e-3 * np.abs(s_ref)) > 0.95
    # deploy: fake_quant_weight_static with round_zp=False on the golden buffers (spqr.py:357-380)
    wq = IntegerQuantizer(cfg.bit, False, 'per_group', group_size=cfg.group_size, round_zp=False)
    w = torch.from_numpy(g[p + 'weight']).cuda()
    mask = torch.from_numpy(g[p + 'buf_mask'].astype(np.float32)).cuda()
    out = (mask * w).to(dt)
    if cfg.actorder:
        perm = torch.from_numpy(g[p + 'perm']).cuda()
        w = w[:, perm]
    args = {'scales': torch.from_numpy(g[p + 'buf_scales']).cuda(), 'zeros': torch.from_numpy(g[p + 'buf_zeros']).cuda(),
            'qmax': torch.tensor(float(2 ** cfg.bit - 1)), 'qmin': torch.tensor(0.0)}
    fq = wq.fake_quant_weight_static(w, args).to(dt)
    if cfg.actorder:
        fq = fq[:, torch.argsort(perm)]
    res = (fq * (1 - mask) + out).to(dt)
    np.testing.assert_array_equal(res.float().cpu().numpy(), g[p + 'w_qdq'])


class Cfg(dict):
    __getattr__ = dict.get


def test_spqr_class_matches_reference_class():
    """llmc_amd's SpQR (ctor -> run_block_loop -> deploy) against the reference's class on the toy adapter
    (tests/golden/e2e_spqr.npz): buffers with the reference's names / shapes / dtypes, statistical agreement of the
    compensated weights (later layers see quantized inputs), refusal of real_quant."""
    import llmc_amd.compression.quantization as Q
    from llmc_amd.compression.quantization.spqr import SpQR
    from toy_model import ToyModel, calib_input
    g = load_golden('e2e_spqr')
    model = ToyModel(hidden=128, inner=256, seed=3)
    config = Cfg(calib=Cfg(seq_len=64), model=Cfg(type='Toy'))
    q2 = Cfg(bit=3, symmetric=False, granularity='per_group', group_size=16, round_zp=False)
    qc = Cfg(weight=Cfg(bit=4, symmetric=False, granularity='per_group', group_size=16, round_zp=False),
             special=Cfg(actorder=True, percdamp=1, blocksize=128, true_sequential=True, relative_threshold=0.2,
                         simplified_outliers=False, scale=Cfg(q2), zero=Cfg(q2)), quant_out=True)
    assert Q.SpQR is SpQR
    algo = SpQR(model, qc, calib_input(model), None, config)
    algo.run_block_loop()
    lin = {f'{i}.{n}': m for i, b in enumerate(model.get_blocks()) for n, m in b.named_modules()
           if hasattr(m, 'weight') and m.weight is not None and m.weight.dim() == 2}
    for n, m in lin.items():
        ref = g[f'w/{n}']
        got = m.weight.data.float().cpu().numpy()
        assert m.weight.dtype == torch.float32 and got.shape == ref.shape
        first = n.startswith('0.gate') or n.startswith('0.up')
        close = np.mean(np.abs(got - ref) < 2e-2 * np.abs(ref).max())
        from conftest import report
        report(f'spqr_vs_reference_class/{n}', w_close_2e2=close, w_close_1e3=np.mean(np.abs(got - ref) < 1e-3 * np.abs(ref).max()))
        # measured 1.0 for every layer of both blocks, also within 1e-3 (gpurun_out/r03c/actuals.jsonl)
        assert close >= 0.999 and np.mean(np.abs(got - ref) < 1e-3 * np.abs(ref).max()) >= 0.995, (n, close)
        assert m.buf_scales.shape == (ref.shape[0] * ref.shape[1] // 16, 1) and m.buf_scales.dtype == torch.float32
        assert m.buf_zeros.shape == m.buf_scales.shape and m.buf_mask.is_sparse
        s_ref = g[f'scales/{n}']
        sg = m.buf_scales.cpu().numpy().reshape(-1)
        s_close = np.mean(np.abs(sg - s_ref) <= 2e-2 * np.abs(s_ref))
        report(f'spqr_vs_reference_class/scales/{n}', s_close_2e2=s_close, s_close_1e4=np.mean(np.abs(sg - s_ref) <= 1e-4 * np.abs(s_ref)))
        assert s_close >= 0.999, (n, s_close)            # measured 1.0 for every layer, also within 1e-4 (profiles/r03_e2e_measured_values.jsonl)
        nout, nref = int(m.buf_mask.to_dense().sum().item()), int(g[f'nout/{n}'])
        assert abs(nout - nref) <= max(4, 0.3 * nref), (n, nout, nref)
    with pytest.raises(AssertionError):
        algo.deploy('real_quant')
    algo.deploy('fake_quant')
    for n in ('0.gate_proj', '1.down_proj'):
        fq = lin[n].weight.data.float().cpu().numpy() if hasattr(lin[n], 'weight') else None
        blk, name = n.split('.')
        mod = getattr(model.get_blocks()[int(blk)], name)
        fq = mod.weight.data.float().cpu().numpy()
        ref = g[f'fake/{n}']
        assert mod.weight.dtype == torch.bfloat16 and fq.shape == ref.shape
        assert np.mean(np.abs(fq - ref) < 0.15 * np.abs(ref).max()) > 0.8
