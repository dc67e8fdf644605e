"""The host classes end to end on a toy two-block model on MI355X: GPTQ / Awq / RTN through run_block_loop and
deploy, checked against the oracle run on the same data."""
import copy

import numpy as np
import pytest
import torch

from oracle import gptq_ref as G
from oracle import quant_ref as Qr

pytestmark = pytest.mark.gpu


class Cfg(dict):
    __getattr__ = dict.get


def make(method_cfg):
    from toy_model import ToyModel, calib_input
    model = ToyModel()
    return model, calib_input(model), Cfg(calib=Cfg(seq_len=64), model=Cfg(type='Toy'))


def test_rtn_deploy_fake_and_vllm_pack_bit_exact():
    import llmc_amd.compression.quantization as Q
    model, inp, config = make(None)
    w0 = {i: {n: m.weight.data.clone() for n, m in model.get_block_linears(b).items()} for i, b in enumerate(model.get_blocks())}
    qc = Cfg(weight=Cfg(bit=4, symmetric=True, granularity='per_group', group_size=128, need_pack=True))
    algo = Q.RTN(model, qc, inp, None, config)
    algo.run_block_loop()
    algo.deploy('vllm_quant')
    blk = model.get_blocks()[0]
    m = blk.gate_proj
    assert type(m).__name__ == 'VllmRealQuantLinear' and m.weight_packed.dtype == torch.int32
    w = w0[0]['gate_proj'].float().numpy()
    codes, s, _ = Qr.real_quant_dynamic(w.reshape(-1, 128), 'bf16', True, -8.0, 7.0)
    np.testing.assert_array_equal(m.weight_packed.cpu().numpy(), Qr.pack_lsb(codes.reshape(w.shape), 4))
    np.testing.assert_array_equal(m.weight_scale.float().cpu().numpy(), Qr.rnd(s.reshape(w.shape[0], -1), 'f16'))


@pytest.mark.parametrize('variant', ['dyn', 'static'])
def test_gptq_block_loop_matches_oracle_pipeline(variant):
    import llmc_amd.compression.quantization as Q
    model, inp, config = make(None)
    ref_model = copy.deepcopy(model)
    static = variant == 'static'
    qc = Cfg(weight=Cfg(bit=4, symmetric=static, granularity='per_group', group_size=128),
             special=Cfg(actorder=True, static_groups=static, percdamp=0.01, blocksize=128, true_sequential=True),
             quant_out=True)
    algo = Q.GPTQ(model, qc, copy.deepcopy(inp), None, config)
    algo.run_block_loop()
    blk = model.get_blocks()[0]
    assert blk.gate_proj.weight.dtype == torch.float32
    assert blk.gate_proj.buf_perm.shape == (256,)
    if not static:
        assert blk.gate_proj.buf_scales.dtype == torch.float32 and blk.gate_proj.buf_scales.shape == (384 * 2, 1)
    # oracle: first block, first subset, from the same calibration activations
    # the block ran in bf16 on the GPU: take the LN output from a bf16 run for an apples-to-apples Hessian
    rb16 = copy.deepcopy(ref_model.get_blocks()[0]).cuda()
    h16 = torch.cat([rb16.ln(x.cuda()) for x in inp['data']], dim=0)
    H = np.zeros((256, 256), np.float32)
    n = 0
    for i in range(h16.shape[0]):
        H, n = G.add_batch(H, n, h16[i].detach().float().cpu().numpy())
    W = ref_model.get_blocks()[0].gate_proj.weight.data.float().numpy()
    perm = G.hessian_sorting(H)
    Wp, U = G.process_hessian_and_weights(W, H, perm, 0.01)
    qmin, qmax = Qr.int_range(4, static)
    if static:
        s0, z0 = Qr.minmax_qparams(W.reshape(-1, 128), 'bf16', True, qmin, qmax)
        r = G.weight_transform(Wp, U, True, qmin, qmax, 128, True, (perm // 128).astype(np.int32),
                               s0.reshape(384, -1), None)
    else:
        r = G.weight_transform(Wp, U, False, qmin, qmax, 128)
    ref_w = r['tmp'][:, np.argsort(perm)]
    got = blk.gate_proj.weight.data.cpu().numpy()
    # different fp32 factorisation order => not bit-identical; the compensated weights agree closely and the
    # layer output error ||X (W' - W'_ref)^T|| is far below the quantisation error itself
    # (GPTQ is chaotic in the last bit: one flipped rounding moves the rest of that row by up to a quantisation
    # step, so the max-norm bound is a step of the 4-bit grid, the bulk of the weights agree to 1e-3)
    rel = np.abs(got - ref_w).max() / np.abs(ref_w).max()
    close3 = np.mean(np.abs(got - ref_w) < 1e-3 * np.abs(ref_w).max())
    from conftest import report
    report('gptq_block_loop_vs_oracle/' + variant, rel_max=rel, close_1e3=close3)
    # measured (gpurun_out/r03c/actuals.jsonl): dynamic groups 3.4e-6 / 1.0, static groups 0.144 / 0.9969 — with static
    # groups a last-bit difference can flip one 4-bit code, which moves that weight by a quantisation step (rel_max) and
    # a few of its row's later weights through the error feedback; nothing flips with dynamic groups on this layer
    assert rel < (0.3 if static else 1e-4), rel
    assert close3 > (0.99 if static else 0.9999), close3
    # the quantity GPTQ minimises: output error of the layer on the calibration activations
    X = h16.detach().reshape(-1, 256).float().cpu().numpy()
    e_got = np.linalg.norm(X @ (got - W).T) / np.linalg.norm(X @ W.T)
    e_ref = np.linalg.norm(X @ (ref_w - W).T) / np.linalg.norm(X @ W.T)
    assert abs(e_got - e_ref) <= 0.1 * e_ref + 1e-4, (e_got, e_ref)
    # deploy fake quant restores the model dtype and applies the stored qparams
    algo.deploy('fake_quant')
    fq = model.get_blocks()[0].gate_proj
    assert type(fq).__name__ == 'EffcientFakeQuantLinear' and fq.weight.dtype == torch.bfloat16
    if static:
        model2, inp2, _ = make(None)
        algo2 = Q.GPTQ(model2, qc, copy.deepcopy(inp2), None, config)
        algo2.run_block_loop()
        algo2.deploy('vllm_quant')
        assert model2.get_blocks()[1].down_proj.weight.dtype == torch.int32


def test_awq_trans_and_clip_run_and_preserve_function():
    import llmc_amd.compression.quantization as Q
    model, inp, config = make(None)
    ref = copy.deepcopy(model)
    qc = Cfg(weight=Cfg(bit=4, symmetric=True, granularity='per_group', group_size=128),
             special=Cfg(trans=True, trans_version='v2', weight_clip=True, clip_sym=True))
    inp1 = {'data': [torch.cat(inp['data'], dim=0)], 'kwargs': [{}]}        # calib.bs = -1: one batch
    algo = Q.Awq(model, qc, inp1, None, config)
    algo.run_block_loop()
    # scale folding keeps the float function (before quantisation): ln.w / s and fc.W * s cancel, up to clipping
    x = torch.cat(inp['data'], dim=0).cuda()
    b_new, b_old = model.get_blocks()[0].cuda(), ref.get_blocks()[0].cuda()
    y_new, y_old = b_new(x).detach().float(), b_old(x).detach().float()
    assert ((y_new - y_old).norm() / y_old.norm()).item() < 0.1
    s = (ref.get_blocks()[0].ln.weight.data.float().cpu() / model.get_blocks()[0].ln.weight.data.float().cpu())
    assert s.min() > 0 and s.max() / s.min() > 1.5                            # a non-trivial scale was applied
    algo.deploy('fake_quant')
    assert type(model.get_blocks()[0].gate_proj).__name__ == 'EffcientFakeQuantLinear'


# ---- the same adapter / config / seeds through the REFERENCE's own classes on CPU: tests/golden/e2e.npz ----------
def _linears(model):
    return {f'{i}.{n}': m for i, b in enumerate(model.get_blocks()) for n, m in b.named_modules()
            if getattr(m, 'weight', None) is not None and torch.is_tensor(m.weight) and m.weight.dim() == 2}


def _toy():
    from toy_model import ToyModel, calib_input
    model = ToyModel(hidden=128, inner=256, seed=3)
    return model, calib_input(model), Cfg(calib=Cfg(seq_len=64), model=Cfg(type='Toy'))


def test_rtn_matches_reference_classes_bit_exact():
    import llmc_amd.compression.quantization as Q
    from conftest import load_golden
    g = load_golden('e2e')
    model, inp, config = _toy()
    qc = Cfg(weight=Cfg(bit=4, symmetric=True, granularity='per_group', group_size=128))
    algo = Q.RTN(model, qc, inp, None, config)
    algo.run_block_loop()
    algo.deploy('fake_quant')
    for n, m in _linears(model).items():
        np.testing.assert_array_equal(m.weight.data.float().cpu().numpy(), g['rtn/' + n], err_msg=n)


@pytest.mark.parametrize('tag,sym,static', [('gptq_dyn', False, False), ('gptq_static', True, True)])
def test_gptq_matches_reference_classes(tag, sym, static):
    """Same block loop (true_sequential + quant_out re-forwards included) as the reference's GPTQ class. The
    factorisation order differs (fp32), GPTQ's error feedback amplifies last-bit differences, and later subsets /
    blocks see inputs produced by earlier quantised layers, so agreement is statistical and degrades with depth."""
    import llmc_amd.compression.quantization as Q
    from conftest import load_golden
    g = load_golden('e2e')
    model, inp, config = _toy()
    qc = Cfg(weight=Cfg(bit=4, symmetric=sym, granularity='per_group', group_size=128),
             special=Cfg(actorder=True, static_groups=static, percdamp=0.01, blocksize=128, true_sequential=True),
             quant_out=True)
    algo = Q.GPTQ(model, qc, inp, None, config)
    algo.run_block_loop()
    for n, m in _linears(model).items():
        got, ref = m.weight.data.float().cpu().numpy(), g[f'{tag}/w/{n}']
        scale = np.abs(ref).max()
        close = np.mean(np.abs(got - ref) < 2e-2 * scale)
        close3 = np.mean(np.abs(got - ref) < 1e-3 * scale)
        first = n.startswith('0.gate') or n.startswith('0.up')
        s_got, s_ref = m.buf_scales.float().cpu().numpy().reshape(-1), g[f'{tag}/scales/{n}']
        assert s_got.shape == s_ref.shape
        s_close = np.mean(np.abs(s_got - s_ref) <= 2e-2 * np.abs(s_ref))
        s_close4 = np.mean(np.abs(s_got - s_ref) <= 1e-4 * np.abs(s_ref))
        from conftest import report
        report(f'gptq_vs_reference_classes/{tag}/{n}', w_close_2e2=close, w_close_1e3=close3, s_close_2e2=s_close,
               s_close_1e4=s_close4)
        # Bounds from the measured values (gpurun_out/r03c/actuals.jsonl) and the envelope (profiles/r03_parity_envelope.txt).
        # Block 0 and every static-groups layer: all weights within 1e-3 and all scales within 1e-4 of the reference's
        # (measured 1.0). Dynamic groups, block 1: its inputs come from block 0's quantised layers (quant_out +
        # true_sequential), so one flipped code upstream changes the calibration data itself: gate / up measured
        # 0.9968 / 0.9983 within 1e-3, down_proj (third generation, K = 256, 384 calibration tokens) 0.92 within 1e-3,
        # 0.975 within 2e-2, scales 0.988 within 2e-2. The reference run twice (host vs ROCm) shows the same kind of
        # spread at Llama width (envelope: codes 0.99912 at K = 14336 with 65536 tokens).
        block0 = n.startswith('0.')
        if block0 or static:
            assert close3 >= 0.999 and s_close4 >= 0.999, (n, close3, s_close4)
        elif n.endswith('down_proj'):
            assert close >= 0.95 and close3 >= 0.85 and s_close >= 0.97, (n, close, close3, s_close)
        else:
            assert close3 >= 0.99 and s_close4 >= 0.999, (n, close3, s_close4)
    algo.deploy('fake_quant')
    fq = model.get_blocks()[0].down_proj.weight.data.float().cpu().numpy()
    ref = g[f'{tag}/fake/0.down_proj']
    assert fq.shape == ref.shape and np.mean(np.abs(fq - ref) < 0.15 * np.abs(ref).max()) > 0.8


def test_awq_matches_reference_classes():
    import llmc_amd.compression.quantization as Q
    from conftest import load_golden
    g = load_golden('e2e')
    model, inp, config = _toy()
    inp1 = {'data': [torch.cat(inp['data'], dim=0)], 'kwargs': [{}]}
    qc = Cfg(weight=Cfg(bit=4, symmetric=True, granularity='per_group', group_size=128),
             special=Cfg(trans=True, trans_version='v2', weight_clip=True, clip_sym=True))
    algo = Q.Awq(model, qc, inp1, None, config)
    algo.run_block_loop()
    # block 0: same calibration input as the reference -> same grid point, scales within a few ulp (bf16)
    ln_got = model.get_blocks()[0].ln.weight.data.float().cpu().numpy()
    ln_ref = g['awq/ln/0']
    assert np.mean(np.abs(ln_got - ln_ref) <= 2.0 ** -6 * np.abs(ln_ref)) > 0.98
    for n in ('0.gate_proj', '0.up_proj'):
        got = _linears(model)[n].weight.data.float().cpu().numpy()
        ref = g['awq/w/' + n]
        assert np.mean(np.abs(got - ref) <= 2.0 ** -5 * np.abs(ref).max()) > 0.97, n


def test_gptq_owq_block_loop_keeps_outlier_columns_in_floating_point():
    """OWQ through the class (quant.special.owq / n_outs, gptq.py:44-56,89-93): buf_n_nonout / buf_perm are set, the
    n_out columns with the largest Hessian diagonal are NOT on the 4-bit grid after deploy('fake_quant'), the rest are."""
    import llmc_amd.compression.quantization as Q
    model, inp, config = make(None)
    qc = Cfg(weight=Cfg(bit=4, symmetric=False, granularity='per_group', group_size=128),
             special=Cfg(actorder=True, static_groups=True, percdamp=0.01, blocksize=128, true_sequential=False,
                         owq=True, n_outs=[6, 6, 8]),
             quant_out=False)
    algo = Q.GPTQ(model, qc, copy.deepcopy(inp), None, config)
    assert algo.owq and not algo.actorder and not algo.static_groups and algo.need_perm      # forced by OWQ
    algo.run_block_loop()
    blk = model.get_blocks()[0]
    for name, n_out in (('gate_proj', 6), ('up_proj', 6), ('down_proj', 8)):
        m = getattr(blk, name)
        K = m.weight.shape[1]
        assert int(m.buf_n_nonout) == K - n_out and m.buf_perm.shape == (K,)
        assert m.weight.dtype == torch.float32
        assert sorted(m.buf_perm.tolist()) == list(range(K))
    algo.deploy('fake_quant')
    fq = model.get_blocks()[0].gate_proj
    w = fq.weight.float()
    perm = fq.buf_perm
    out_cols = perm[-6:]
    # quantized columns: at most 16 distinct values per (row, group of the PERMUTED order); outlier columns are free
    wp = w[:, perm][:, :256 - 6]
    g0 = wp[:, :128]
    assert max(len(torch.unique(g0[r])) for r in range(8)) <= 16
    assert len(torch.unique(w[:, out_cols])) > 16 * 6
