"""End-to-end GPTQ parity at a Llama input width against the REFERENCE'S OWN CLASS METHODS run on the host cores
(oracle/_ref = llmc after its own CI rewrite; driven by tools/parity_arm.py exactly as tools/parity_envelope.py does).
The bounds come from the measured envelope, profiles/r03_parity_envelope.txt: at K = 4096 with the full 128 x 2048
calibration set the reference on the host and the unmodified reference on this GPU (PyTorch-ROCm) agree with each other
on 99.999 % of the INT4 codes, and llmc_amd agrees with either of them on the same 99.999 %; with fewer samples per
channel (this test: 24 x 2048 tokens for 4096 channels, to stay in seconds) the Hessian is worse conditioned and every
pair drops together. Asserted here: what north_star states for the pieces that are comparable across implementations —
static-group scales identical; dynamic-group scales (min / max of the error-compensated weights: a last-bit difference
upstream moves them at the 1e-4 .. 1e-3 level in the reference's own two runs as well, scale_rel_max 2.4e-3 in the envelope)
within 1e-4 on >= 99 % and within 1e-2 on >= 99.9 % of the groups; INT4 codes identical on >= 99.95 % of the weights;
zero points identical; the same actorder permutation up to tied diagonals; the same layer-output error and sum(Losses).
Measured on the round-3 box (gpurun_out/r03c/actuals.jsonl): codes 0.99993, scales within 1e-4 0.9953, zeros 1.0,
perm 0.9995."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'llmc')),
                    reason='oracle/_ref (the reference, built by __graft_entry__.build() in the build container) is absent')
def test_gptq_layer_matches_the_reference_class_within_the_measured_envelope(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import parity_envelope as PE
    from conftest import report
    dev = torch.device('cuda', 0)
    R, K, n_seq, seq = 512, 4096, 24, 2048
    W, X = PE.synth(R, K, n_seq, seq, 5, dev)
    data = str(tmp_path / 'data.pt')
    torch.save({'W': W.cpu(), 'X': X.cpu()}, data)
    res = {}
    for arm, extra in (('ref_cpu', ['--threads', '16']), ('ours', []), ('ours_fp32diag', [])):
        out = str(tmp_path / f'{arm}.npz')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'parity_arm.py'), '--arm', arm, '--data', data, '--out', out] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (arm, r.stderr[-800:])
        res[arm] = dict(np.load(out))
    Xs = X[:4].reshape(-1, K).float()
    Y0 = Xs @ W.float().T
    for v in ('w_only', 'vllm'):
        qa, sa, za, wa = PE.codes_of(res['ref_cpu'], v, K)
        qb, sb, zb, wb = PE.codes_of(res['ours'], v, K)
        codes = float((qa == qb).float().mean())
        rel_s = (sa - sb).abs() / sb.abs().clamp_min(1e-30)
        s_ok = float((rel_s <= 1e-4).float().mean())
        z_eq = float((za == zb).float().mean())
        perm = float((torch.from_numpy(res['ref_cpu'][v + '/perm']) == torch.from_numpy(res['ours'][v + '/perm'])).float().mean())
        ea = float(((Xs @ (wa.to(dev) - W.float()).T).double() ** 2).sum() / (Y0.double() ** 2).sum())
        eb = float(((Xs @ (wb.to(dev) - W.float()).T).double() ** 2).sum() / (Y0.double() ** 2).sum())
        la, lb = float(res['ref_cpu'][v + '/loss']), float(res['ours'][v + '/loss'])
        s_ok2 = float((rel_s <= 1e-2).float().mean())
        report('envelope_llama_width/' + v, codes_equal=codes, scales_within_1e4=s_ok, scales_within_1e2=s_ok2, zeros_equal=z_eq,
               perm_equal=perm, out_err_ref=ea, out_err_ours=eb, loss_ref=la, loss_ours=lb)
        assert codes >= 0.9995, (v, codes)
        assert s_ok >= 0.99 and s_ok2 >= 0.999 and z_eq >= 0.999, (v, s_ok, s_ok2, z_eq)
        if v == 'vllm':
            assert float(rel_s.max()) == 0.0                       # static groups: RTN scales of the original weights
        assert perm >= 0.995, (v, perm)
        assert abs(ea - eb) <= 1e-3 * ea and abs(la - lb) <= 1e-4 * abs(la), (v, ea, eb, la, lb)
    # round 6: diag(H) is folded into fp64 by the MFMA kernel itself (the DEFAULT: hessian_syrk.hip, exact diagonal), so the sort
    # key of actorder carries the noise of the reference's own sgemm, not twice that (profiles/r05_parity_envelope_full_down.txt:
    # 1.18e-6 against 3.07e-6 for the fp32 chain at the bench's down_proj, where the reference on the host and on ROCm differ by
    # 1.65e-6), and every agreement figure of the default arm reaches the reference-vs-itself value. `ours_fp32diag` is the A/B arm
    # (HessianAccumulator(exact_diag=False): the MFMA kernel's own fp32 diagonal, rounds 1-5's default).
    dref = res['ref_cpu']['H_diag'].astype(np.float64)
    e_def = float((np.abs(res['ours']['H_diag'] - dref) / dref).max())
    e_f32 = float((np.abs(res['ours_fp32diag']['H_diag'] - dref) / dref).max())
    p_def = float((torch.from_numpy(res['ref_cpu']['w_only/perm']) == torch.from_numpy(res['ours']['w_only/perm'])).float().mean())
    p_f32 = float((torch.from_numpy(res['ref_cpu']['w_only/perm']) == torch.from_numpy(res['ours_fp32diag']['w_only/perm'])).float().mean())
    qa, sa, za, wa = PE.codes_of(res['ref_cpu'], 'w_only', K)
    qx, sx, zx, wx = PE.codes_of(res['ours'], 'w_only', K)
    c_def = float((qa == qx).float().mean())
    s_def = float((((sa - sx).abs() / sx.abs().clamp_min(1e-30)) <= 1e-4).float().mean())
    report('envelope_llama_width/exact_diag', diag_rel_max_default=e_def, diag_rel_max_fp32diag=e_f32, perm_default=p_def,
           perm_fp32diag=p_f32, codes_default=c_def, scales_within_1e4_default=s_def)
    assert e_def <= 1.7e-6 and e_def <= e_f32, (e_def, e_f32)
    assert p_def >= 0.995 and p_def >= p_f32 - 2e-3, (p_def, p_f32)
    assert c_def >= 0.9995 and s_def >= 0.99, (c_def, s_def)
