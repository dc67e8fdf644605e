"""AWQ kernels (K8/K9) on MI355X vs the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import awq_ref as A
from oracle import quant_ref as Q

pytestmark = pytest.mark.gpu
TD = {'f16': torch.float16, 'bf16': torch.bfloat16}


def dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(TD[dt]).cuda()


def host(t):
    return t.detach().float().cpu().numpy()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ulps(a, b, dt):
    sh = 16 if dt == 'bf16' else 13
    a = np.ascontiguousarray(a, dtype=np.float32).ravel().view(np.int32) >> sh
    b = np.ascontiguousarray(b, dtype=np.float32).ravel().view(np.int32) >> sh
    return np.abs(a - b)


def make_q(sym, gs):
    from llmc_amd.compression.quantization import IntegerQuantizer
    return IntegerQuantizer(4, bool(sym), 'per_group', group_size=gs)


def test_elementwise_chain_vs_reference_golden():
    from llmc_amd.compression.quantization import awq_ops
    g = load_golden('awq')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, nl, K = [int(v) for v in g[p + 'meta']]
        dt, ver = str(g[p + 'dt']), str(g[p + 'ver'])
        q = make_q(sym, gs)
        s = dev(g[p + 'scales_r035'], dt)
        wcat = torch.cat([dev(g[p + f'w{i}'], dt) for i in range(nl)], dim=0)
        wq = awq_ops.scale_fakequant(wcat, s, q)
        np.testing.assert_array_equal(bits(host(wq)), bits(g[p + 'wq_r035']), err_msg=name)
        xs = awq_ops.div_cols(dev(g[p + 'x'], dt), s)
        np.testing.assert_array_equal(bits(host(xs)), bits(g[p + 'xs_r035']), err_msg=name)
        xm = awq_ops.act_mean(dev(g[p + 'x'], dt))
        assert ulps(host(xm), g[p + 'x_mean'], dt).max() <= 1, name
        assert (ulps(host(xm), g[p + 'x_mean'], dt) > 0).mean() <= 0.02, name
        wm = None
        for i in range(nl):
            m = awq_ops.weight_mean(dev(g[p + f'w{i}'], dt), gs)
            wm = m if wm is None else wm.add_(m)
        wm = wm.div_(nl)
        assert ulps(host(wm), g[p + 'w_max'], dt).max() <= 2, name
        for tag, ratio in (('scales_r050', 0.5), ('scales_r035', 0.35)):
            sc = awq_ops.awq_scales(dev(g[p + 'x_mean'], dt), dev(g[p + 'w_max'], dt), ratio, ver)
            u = ulps(host(sc), g[p + tag], dt)
            assert u.max() <= 1 and (u > 0).mean() <= 0.03, (name, tag, u.max(), (u > 0).mean())


def test_search_matches_reference_golden():
    from llmc_amd.compression.quantization.awq_pipeline import search_scale_stacked
    g = load_golden('awq')
    for name in [str(n) for n in g['names']]:
        p = name + '/'
        sym, gs, nl, K = [int(v) for v in g[p + 'meta']]
        dt, ver = str(g[p + 'dt']), str(g[p + 'ver'])
        q = make_q(sym, gs)
        ws = [dev(g[p + f'w{i}'], dt) for i in range(nl)]
        w_before = [w.clone() for w in ws]
        best, losses, n = search_scale_stacked(ws, dev(g[p + 'x'], dt), q, ver, return_losses=True)
        for w, w0 in zip(ws, w_before):
            assert torch.equal(w, w0)
        ref = g[p + 'losses']
        np.testing.assert_allclose(losses.cpu().numpy(), ref, rtol=2e-2, err_msg=name)
        assert n == int(np.argmin(ref)), name
        u = ulps(host(best), g[p + 'best_scales'], dt)
        assert u.max() <= 2, (name, u.max())


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('shape', [(512, 256, 256), (1000, 512, 320), (4096, 4096, 1024)])
def test_linear_eval_vs_fp32(dt, shape):
    from llmc_amd.compression.quantization import awq_ops
    N, K, R = shape
    gen = torch.Generator().manual_seed(N + R)
    x = torch.randn(N, K, generator=gen).to(TD[dt]).cuda()
    w = (torch.randn(R, K, generator=gen) * 0.05).to(TD[dt]).cuda()
    y = awq_ops.linear_out(x, w)
    ref32 = x.float() @ w.float().T
    ref = ref32.to(TD[dt])
    # the result is the fp32 sum rounded once to the model dtype: at most one ulp from torch's own rounding of
    # its fp32 GEMM, except next to zero where cancellation makes ulps meaningless -> absolute floor
    eps = 2.0 ** -7 if dt == "bf16" else 2.0 ** -10
    err = (y.float() - ref.float()).abs()
    tol = eps * ref.float().abs() + 1e-5 * ref32.abs().max()
    assert bool((err <= tol).all())
    assert (ulps(host(y), host(ref), dt) > 0).mean() < 0.01
    # asymmetric pattern (transposition check): one hot row / column
    y0 = (ref32 * 0.9).to(TD[dt])
    loss = awq_ops.linear_loss_sum(x, w, y0)
    d = (y0.float() - y.float()).to(TD[dt]).float()
    expect = (d * d).sum().item()
    assert abs(loss.item() - expect) / expect < 1e-4
