"""Parity at the shapes bench.py publishes numbers on besides the headline ("untested published numbers"):
BASELINE configs[3] — GPTQ on Llama-3-70B shapes (K = 8192 for q|k|v / o / gate|up, K = 28672 for down_proj; stacked gate|up
R = 57344) — and configs[2] — the AWQ scale search at N = 65536 tokens on Llama-3-8B shapes, plus the row-chunked search on a
stack whose reference output really exceeds 4 GiB (70B gate|up). Reference ops: gptq.py:128-244, 254-295; awq.py:179-278."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import gptq_ref as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def spd(K, seed, damp=0.05):
    """(X^T X) / K + damp * I with X [2K, K] standard normal, accumulated in two halves (fp32, on the device)."""
    g = torch.Generator(device='cuda').manual_seed(seed)
    H = torch.zeros((K, K), device='cuda')
    for _ in range(2):
        X = torch.randn((K, K), generator=g, device='cuda')
        H.addmm_(X.T, X, alpha=1.0 / K)
        del X
    H.diagonal().add_(damp)
    return H


# ---- K1 at the 70B launches -------------------------------------------------------------------------------------------

def test_hessian_70b_k8192_bench_launch():
    """q|k|v, o, gate|up of Llama-3-70B: 128 x 2048 tokens, K = 8192, bf16 — exactly bench.py --model llama3-70b's launch."""
    from test_bench_shapes_gpu import check_tiles, run_hessian, synth_x
    T, K, n_seq = 262144, 8192, 128
    x = synth_x(T, K, torch.bfloat16, 11)
    H = run_hessian(x, n_seq)
    tiles = [(0, 0), (31, 31), (31, 0), (31, 30), (16, 15), (20, 3), (9, 9), (27, 12), (1, 0)]
    check_tiles(H, x, n_seq, tiles)
    assert torch.isfinite(H).all() and torch.equal(H, H.T)


def test_hessian_70b_down_k28672_bench_launch():
    """down_proj of Llama-3-70B: K = 28672 (112 x 112 tiles, 14 GiB of activations, a 3.1-GiB Hessian), the full 128 x 2048
    tokens; sampled tiles (diagonal, first / last row and column, interior) against fp64."""
    from test_bench_shapes_gpu import check_tiles, run_hessian, synth_x
    T, K, n_seq = 262144, 28672, 128
    x = synth_x(T, K, torch.bfloat16, 12)
    H = run_hessian(x, n_seq)
    tiles = [(0, 0), (111, 111), (111, 0), (111, 110), (56, 55), (77, 13), (100, 100), (64, 63), (1, 0)]
    check_tiles(H, x, n_seq, tiles)
    del x
    assert torch.isfinite(H.diagonal()).all()
    for r0 in range(0, K, 4096):            # full symmetry, slab by slab (no 3-GiB temporaries)
        assert torch.equal(H[r0:r0 + 4096], H[:, r0:r0 + 4096].T)


def test_hessian_70b_down_per_sample_table_equals_one_tensor():
    """calib.bs = 1 at K = 28672: 128 separately allocated [1, 2048, K] tensors through the sample table (one launch) give the
    bits of the one-tensor launch (a quarter of the tokens: the property does not depend on the sample count)."""
    from llmc_amd.compression.quantization.hessian import HessianAccumulator
    from test_bench_shapes_gpu import run_hessian, synth_x
    T, K, n_seq = 65536, 28672, 32
    x = synth_x(T, K, torch.bfloat16, 13)
    H1 = run_hessian(x, n_seq).clone()
    seq = T // n_seq
    samples = [x[i * seq:(i + 1) * seq].clone().unsqueeze(0) for i in range(n_seq)]
    del x
    acc = HessianAccumulator(K, 'cuda')
    acc.timing = []
    for s in samples:
        acc.add(s)
    H2 = acc.H
    assert acc.nsamples == n_seq and len(acc.timing) == 1
    assert torch.equal(H1, H2)


# ---- K2 / K3 / K4 at K = 28672 ------------------------------------------------------------------------------------------

def test_hessian_prep_and_gather_cols_k28672():
    """llmc_hessian_prep / llmc_gather_cols with a 112-KB row staged in LDS (K > 16384 used to leave the HIP path:
    gptq_pipeline.py fell back to index_select). Bit-exact against plain indexing."""
    from llmc_amd.compression.quantization.gptq_ops import gather_cols, hessian_prep
    K, R = 28672, 320
    g = torch.Generator(device='cuda').manual_seed(3)
    H = torch.randn((K, K), generator=g, device='cuda')
    H = H + H.T
    d = torch.rand(K, generator=g, device='cuda') + 1.0
    dead = torch.randperm(K, generator=g, device='cuda')[:5]
    d[dead] = 0.0
    H.diagonal().copy_(d)
    W = (torch.randn((R, K), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
    perm = torch.argsort(torch.diagonal(H), descending=True)
    H0 = H.clone()
    Hout, Wout = hessian_prep(H, W, perm, 0.01)
    dfix = d.clone()
    dfix[dead] = 1.0
    assert torch.equal(H.diagonal(), dfix)                                   # in-place dead fix (gptq.py:139-141)
    damp = 0.01 * dfix.double().mean()
    for r0 in range(0, K, 2048):                                             # slabs: no second 3-GiB temporary
        ref = H0[perm[r0:r0 + 2048]][:, perm]
        got = Hout[r0:r0 + 2048].clone()
        idx = torch.arange(r0, min(K, r0 + 2048), device='cuda')
        dg = got[idx - r0, idx].double()
        want = dfix[perm[idx]].double() + damp
        assert ((dg - want).abs() <= 1e-6 * want.abs()).all()
        got[idx - r0, idx] = 0
        ref[idx - r0, idx] = 0
        assert torch.equal(got, ref)
    Wref = W.float()[:, perm]
    Wref[:, torch.isin(perm, dead)] = 0.0
    assert torch.equal(Wout, Wref)
    inv = torch.argsort(perm)
    assert torch.equal(gather_cols(Wout, inv), Wout[:, inv])
    big = torch.randn((2048, K), generator=g, device='cuda')
    assert torch.equal(gather_cols(big, inv), big[:, inv])


@pytest.fixture(scope='module')
def factor_k28672():
    from llmc_amd.compression.quantization.gptq_ops import chol_inv_upper
    K = 28672
    H = spd(K, 28672)
    U, info = chol_inv_upper(H.clone(), check=False, return_info=True)
    assert int(info.item()) == 0
    return H, U


def test_chol_inv_upper_k28672_property(factor_k28672):
    """The factor the 70B down_proj needs: U upper, U H U^T = I (fp32 on the GPU). 224 factor steps, doubling levels up to
    16384 + 12288 (uneven), 3.1-GiB matrices."""
    from conftest import report
    H, U = factor_k28672
    K = H.shape[0]
    worst = 0.0
    for r0 in range(0, K, 4096):
        Us = U[r0:r0 + 4096]
        assert torch.equal(Us, torch.triu(Us, diagonal=r0))                  # upper triangular, slab by slab
        E = (Us @ H) @ U.T                                                   # rows r0 .. of U H U^T
        idx = torch.arange(Us.shape[0], device='cuda')
        E[idx, idx + r0] -= 1.0
        worst = max(worst, E.abs().max().item())
        del E
    report('chol_inv_upper_k28672', max_abs_UHUt_minus_I=worst)
    assert worst < 3e-3, worst


def test_column_loop_k28672_sampled_rows_bit_exact(factor_k28672):
    """K4 at the 70B down_proj width (224 blocks, far updates up to 28160 columns wide) on the factor K3 produced: four
    sampled rows against oracle/csrc/gptq_canon.c run on those rows (rows are independent given Hinv), bit for bit."""
    from llmc_amd.compression.quantization.gptq_ops import gptq_quantize
    _, U = factor_k28672
    K, R = U.shape[0], 1024
    g = torch.Generator(device='cuda').manual_seed(5)
    W = torch.randn((R, K), generator=g, device='cuda') * 0.02
    W[:, torch.randperm(K, generator=g, device='cuda')[:28]] *= 20.0
    W0 = W.clone()
    tmp, losses, s, z = gptq_quantize(W, U, False, 0.0, 15.0, 128)
    rows = [0, 517, 64, 1023]
    Uh = U.cpu().numpy()
    ref = G.weight_transform(W0[rows].cpu().numpy(), Uh, False, 0.0, 15.0, 128, False, None, None, None)
    np.testing.assert_array_equal(bits(tmp[rows].cpu().numpy()), bits(ref['tmp']))
    np.testing.assert_array_equal(bits(losses[rows].cpu().numpy()), bits(ref['losses']))
    np.testing.assert_array_equal(bits(s[rows].cpu().numpy()), bits(ref['scales']))
    np.testing.assert_array_equal(z[rows].cpu().numpy(), ref['zeros'])
    assert torch.isfinite(tmp).all()


def test_quantize_stacked_70b_down_shape_runs_on_the_hip_path(factor_k28672, monkeypatch):
    """quantize_stacked at K = 28672 end to end (prep, factor, loop, un-permutation): the un-permutation goes through
    llmc_gather_cols (no ATen index_select), and equals indexing the loop's output."""
    from llmc_amd.compression.quantization import gptq_ops
    from llmc_amd.compression.quantization.gptq_pipeline import GptqConfig, quantize_stacked
    H, _ = factor_k28672
    K, R = H.shape[0], 256
    calls = []
    real = gptq_ops.gather_cols
    monkeypatch.setattr(gptq_ops, 'gather_cols', lambda t, i: (calls.append(tuple(t.shape)), real(t, i))[1])
    g = torch.Generator(device='cuda').manual_seed(6)
    W = (torch.randn((R, K), generator=g, device='cuda') * 0.02).to(torch.bfloat16)
    d = torch.sqrt(torch.rand(K, generator=g, device='cuda') + 0.5)
    Hc = H * d[:, None] * d[None, :]                     # D H D: still SPD, distinct diagonals -> a real actorder permutation
    Href = Hc.clone()
    r = quantize_stacked([W], Hc, GptqConfig(bit=4, symmetric=False, group_size=128, actorder=True, static_groups=False))[0]
    assert int(r.info.item()) == 0 and calls == [(R, K)]
    assert torch.isfinite(r.weight).all() and r.weight.shape == (R, K) and r.scales.shape == (R, K // 128)
    assert not torch.equal(r.perm, torch.arange(K, device='cuda'))
    # what GPTQ is for: the Hessian-weighted output error of the result, quantized with ITS scales / zeros in processing
    # order, is below plain round-to-nearest's on the same weights
    from llmc_amd.compression.quantization import IntegerQuantizer
    wq = IntegerQuantizer(4, False, 'per_group', group_size=128)
    Wf = W.float()
    Wp = r.weight[:, r.perm]
    s = r.scales.reshape(R, -1, 1)
    z = r.zeros.reshape(R, -1, 1)
    q = torch.clamp(torch.round(Wp.reshape(R, -1, 128) / s) + z, 0, 15)
    Wq = torch.empty_like(Wf)
    Wq[:, r.perm] = ((q - z) * s).reshape(R, K)
    rtn = wq.fake_quant_weight_dynamic(W).float()

    def herr(d):
        return float(((d @ Href) * d).sum() / ((Wf @ Href) * Wf).sum())
    e_gptq, e_rtn = herr(Wq - Wf), herr(rtn - Wf)
    from conftest import report
    report('quantize_stacked_k28672', hessian_weighted_err_gptq=e_gptq, hessian_weighted_err_rtn=e_rtn)
    assert e_gptq < e_rtn and e_gptq < 0.05, (e_gptq, e_rtn)


# ---- AWQ at configs[2] ---------------------------------------------------------------------------------------------------

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref_gpu', 'llmc')),
                               reason='oracle/_ref_gpu (plain copy of the reference, made by __graft_entry__.build()) is absent')


def awq_arms(tmp_path, rows, K, N, seed):
    res = {}
    for arm in ('ref_rocm', 'ours'):
        out = str(tmp_path / f'{arm}.npz')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'parity_awq_arm.py'), '--arm', arm, '--rows', rows,
                            '--K', str(K), '--N', str(N), '--seed', str(seed), '--out', out],
                           capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, (arm, r.stdout[-500:], r.stderr[-1500:])
        res[arm] = dict(np.load(out))
    return res


def check_awq(tag, res):
    from conftest import report
    a, b = res['ref_rocm'], res['ours']
    la, lb = a['losses'], b['losses']
    assert la.shape == lb.shape == (20,)
    rel = float(np.abs(la - lb).max() / la.min())
    srt = np.sort(la)
    gap = float((srt[1] - srt[0]) / srt[0])
    xm = float((a['x_mean'] == b['x_mean']).mean())
    wm = float((a['w_max'] == b['w_max']).mean())
    eq = float((a['best'] == b['best']).mean())
    report('awq_config2/' + tag, loss_max_err_over_min=rel, gap_best_second=gap, argmin_ref=int(a['argmin']), argmin_ours=int(b['argmin']),
           x_mean_equal=xm, w_max_equal=wm, best_scales_equal=eq, t_ref=float(a['t_total']), t_ours=float(b['t_total']))
    # both arms accumulate in fp32 and round the outputs to bf16; only the summation order of a 4096- / 8192-deep dot product
    # differs, so the mean-square losses agree to ~1e-4
    assert rel <= 1e-3, (la, lb)
    if rel < gap / 2:
        assert int(a['argmin']) == int(b['argmin'])
    else:
        assert la[int(b['argmin'])] <= srt[1] * (1 + 1e-9)
    assert wm == 1.0                                   # get_weight_scale: element-wise chain, bit-exact
    assert xm >= 0.99                                  # token mean over 65536 tokens: within an ulp of the model dtype
    if int(a['argmin']) == int(b['argmin']):
        assert eq >= 0.99 and float(np.abs(a['best'] - b['best']).max() / np.abs(a['best']).max()) <= 2.0 ** -6
    return rel


@needs_ref
def test_awq_search_at_config2_o_proj_shape_matches_the_reference_on_rocm(tmp_path):
    """o_proj of Llama-3-8B with the shipped calibration size: N = 128 x 512 = 65536 tokens in one batch, K = R = 4096, W4 sym
    g128, trans v2 (configs/quantization/methods/Awq/awq_w_only.yml): the unmodified reference's Awq.search_scale_subset on
    this GPU against search_scale_stacked."""
    res = awq_arms(tmp_path, '4096', 4096, 65536, 21)
    check_awq('o_proj_4096x4096_N65536', res)
    assert int(res['ours']['row_chunked']) == 0


@needs_ref
def test_awq_search_row_chunked_above_4gib_matches_the_reference_on_rocm(tmp_path):
    """The 70B gate|up stack at 65536 tokens: the [N, R] reference output is 65536 x 57344 x 2 B = 7 GiB, so the search walks
    the output rows in chunks (awq_pipeline.py) — here for real, not through the test hook — and must still land on the
    reference's losses and grid point."""
    res = awq_arms(tmp_path, '28672,28672', 8192, 65536, 22)
    check_awq('gate_up_70b_57344x8192_N65536', res)
    assert int(res['ours']['row_chunked']) == 1
