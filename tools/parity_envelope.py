"""End-to-end GPTQ parity ENVELOPE: how far apart are the reference on the host cores,
the unmodified reference on this GPU through PyTorch-ROCm, and llmc_amd, on IDENTICAL weights and calibration samples —
and is llmc_amd inside the spread the reference already has with itself?

    python tools/parity_envelope.py --out gpurun_out/envelope [--quick]

Per layer shape: synthetic W / X, every arm runs tools/parity_arm.py as its own process (separate BLAS /
thread settings), then for every PAIR of arms and both bench variants:
    codes_equal     fraction of identical INT4 codes  (codes = clamp(round(W'/s) + z), computed here with one formula from
                    each arm's own W', scales, zeros — for dynamic groups in the arm's own processing order mapped back to
                    original columns; a column whose group differs because the permutations differ simply counts as its codes do)
    scale_rel_max / scale_rel_med   relative difference of the group scales (dynamic groups; compared per original column)
    zeros_equal     fraction of identical zero points (dynamic groups, asymmetric)
    perm_equal      fraction of positions where the actorder permutations agree
    w_rel_med / w_rel_max   |W'_a - W'_b| / rms(W')
and per arm: sum(Losses), layer output error  ||X (W_hat - W)^T||^2 / ||X W^T||^2  (W_hat = dequantised codes).
Output: <out>.json (all numbers) and <out>.txt (the table committed under profiles/). Measurement infrastructure."""
import argparse
import itertools
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth(R, K, n_seq, seq, seed, dev):
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    w = torch.randn(R, K, generator=g, device=dev, dtype=torch.float32) * 0.02
    idx = torch.randperm(K, generator=g, device=dev)[:max(1, K // 1000)]
    w[:, idx] *= 20.0
    g = torch.Generator(device=dev).manual_seed(2000 + seed)
    c = torch.exp(0.5 * torch.randn(K, generator=g, device=dev))
    c[torch.randperm(K, generator=g, device=dev)[:8]] *= 100.0
    x = torch.empty((n_seq, seq, K), device=dev, dtype=torch.bfloat16)
    for i in range(n_seq):
        x[i] = (torch.randn((seq, K), generator=g, device=dev) * c).to(torch.bfloat16)
    return w.to(torch.bfloat16), x


def codes_of(res, v, K):
    """INT4 codes per ORIGINAL column + per-column scale / zero, from an arm's outputs."""
    W = torch.from_numpy(res[v + '/W'])
    perm = torch.from_numpy(res[v + '/perm']).long()
    s, z = torch.from_numpy(res[v + '/scales']), torch.from_numpy(res[v + '/zeros'])
    R = W.shape[0]
    if v == 'w_only':       # dynamic groups live in processing (permuted) order: column perm[j] belongs to group j // 128
        grp = torch.empty(K, dtype=torch.long)
        grp[perm] = torch.arange(K) // 128
        qmin, qmax = 0.0, 15.0
    else:                   # static groups: original order
        grp = torch.arange(K) // 128
        qmin, qmax = -8.0, 7.0
    sc, zc = s[:, grp], z[:, grp]
    q = torch.clamp(torch.round(W / sc) + zc, qmin, qmax)
    return q.to(torch.int8), sc, zc, (q - zc) * sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/envelope')
    ap.add_argument('--quick', action='store_true', help='tiny shapes (plumbing check)')
    ap.add_argument('--tmp', default='/tmp/llmc_envelope')
    ap.add_argument('--full-down', action='store_true',
                    help='only down_proj 4096 x 14336 with the FULL 128 x 2048 calibration set (the configuration the metric is quoted on)')
    ap.add_argument('--down-70b', action='store_true',
                    help='only the Llama-3-70B down_proj 8192 x 28672 with the full 128 x 2048 calibration set (BASELINE configs[3]): the '
                         'reference on this GPU (ROCm), llmc_amd, and its fp32-diagonal A/B arm (the host arm of the reference needs '
                         '> 15 minutes here and is left out)')
    a = ap.parse_args()
    os.makedirs(a.tmp, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    dev = torch.device('cuda', 0)
    cores = os.cpu_count() or 1
    if a.quick:
        shapes = [('tiny 256x512', 256, 512, 8, 256, [('ref_cpu', 4), ('ref_cpu', 8), ('ref_rocm', 0), ('ours', 0)])]
    else:
        shapes = [
            ('q_proj 4096x4096, 128x2048 tokens', 4096, 4096, 128, 2048,
             [('ref_cpu', 16), ('ref_cpu', 32), ('ref_rocm', 0), ('ours', 0)]),
            ('down_proj 4096x14336, 32x2048 tokens', 4096, 14336, 32, 2048,
             [('ref_cpu', 32), ('ref_rocm', 0), ('ours', 0)]),
        ]
    if a.full_down:
        shapes = [('down_proj 4096x14336, 128x2048 tokens (the bench configuration)', 4096, 14336, 128, 2048,
                   [('ref_cpu', 32), ('ref_rocm', 0), ('ours', 0), ('ours_fp32diag', 0)])]
    if a.down_70b:
        shapes = [('down_proj 8192x28672, 128x2048 tokens (Llama-3-70B, BASELINE configs[3])', 8192, 28672, 128, 2048,
                   [('ref_rocm', 0), ('ours', 0), ('ours_fp32diag', 0)])]
    report = {'cores': cores, 'torch': torch.__version__, 'shapes': []}
    lines = []
    for si, (title, R, K, n_seq, seq, arms) in enumerate(shapes):
        W, X = synth(R, K, n_seq, seq, si, dev)
        data = os.path.join(a.tmp, f'data{si}.pt')
        torch.save({'W': W.cpu(), 'X': X.cpu()}, data)
        Xs = X[: max(1, 8192 // seq)].reshape(-1, K).float()
        Y0 = Xs @ W.float().T
        y0n = float((Y0.double() ** 2).sum().item())
        res = {}
        for arm, thr in arms:
            key = arm + (f'_{thr}t' if thr else '')
            outp = os.path.join(a.tmp, f'res{si}_{key}.npz')
            cmd = [sys.executable, os.path.join(ROOT, 'tools', 'parity_arm.py'), '--arm', arm, '--data', data, '--out', outp]
            if thr:
                cmd += ['--threads', str(thr)]
            t0 = time.time()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500 if (a.full_down or a.down_70b) else 420)
            except subprocess.TimeoutExpired:
                lines.append(f'[{title}] arm {key} TIMED OUT')
                continue
            if r.returncode != 0:
                lines.append(f'[{title}] arm {key} FAILED: {(r.stderr or r.stdout)[-600:]}')
                continue
            res[key] = dict(np.load(outp))
            print(f'[{title}] {key}: {time.time() - t0:.1f} s', flush=True)
        srep = {'title': title, 'R': R, 'K': K, 'n_seq': n_seq, 'seq': seq, 'arms': {}, 'pairs': {}}
        lines.append(f'== {title} ==')
        # Hessian agreement (diag + corner)
        for k, rk in res.items():
            srep['arms'][k] = {'t_hessian_s': float(rk['t_hessian']), 'threads': int(rk['threads'])}
        for v in ('w_only', 'vllm'):
            per = {}
            for k, rk in res.items():
                q, sc, zc, wh = codes_of(rk, v, K)
                e = Xs @ (wh.to(dev) - W.float()).T
                per[k] = (q, sc, zc)
                srep['arms'][k][v] = {'loss_sum': float(rk[v + '/loss']), 't_transform_s': float(rk[v + '/t_transform']),
                                      'layer_out_err': float((e.double() ** 2).sum().item()) / y0n,
                                      'codes_at_clamp': float(((q == (0 if v == 'w_only' else -8)) | (q == (15 if v == 'w_only' else 7))).float().mean())}
            lines.append(f'-- variant {v}: per arm --')
            lines.append(f'{"arm":<16}{"sum(Losses)":>16}{"layer_out_err":>16}{"t_transform[s]":>16}')
            for k in res:
                m = srep['arms'][k][v]
                lines.append(f'{k:<16}{m["loss_sum"]:>16.6g}{m["layer_out_err"]:>16.6g}{m["t_transform_s"]:>16.2f}')
            lines.append(f'-- variant {v}: pairs --')
            lines.append(f'{"pair":<28}{"codes_equal":>12}{"scale_rel_med":>15}{"scale_rel_max":>15}{"zeros_equal":>12}'
                         f'{"perm_equal":>11}{"w_rel_med":>11}{"w_rel_max":>11}{"H_rel_max":>11}')
            for ka, kb in itertools.combinations(res.keys(), 2):
                qa, sa, za = per[ka]
                qb, sb, zb = per[kb]
                Wa, Wb = torch.from_numpy(res[ka][v + '/W']), torch.from_numpy(res[kb][v + '/W'])
                rms = float(Wa.double().pow(2).mean().sqrt())
                dW = (Wa - Wb).abs() / rms
                rel_s = ((sa - sb).abs() / sb.abs().clamp_min(1e-30))
                ha, hb = res[ka]['H_corner'].astype(np.float64), res[kb]['H_corner'].astype(np.float64)
                dd = np.sqrt(np.outer(np.diag(hb), np.diag(hb))) + 1e-300
                m = {'codes_equal': float((qa == qb).float().mean()),
                     'scale_rel_med': float(rel_s.median()), 'scale_rel_max': float(rel_s.max()),
                     'zeros_equal': float((za == zb).float().mean()),
                     'perm_equal': float((torch.from_numpy(res[ka][v + '/perm']) == torch.from_numpy(res[kb][v + '/perm'])).float().mean()),
                     'w_rel_med': float(dW.median()), 'w_rel_max': float(dW.max()),
                     'H_rel_max': float((np.abs(ha - hb) / dd).max())}
                # Where two arms disagree on the actorder permutation, is it a TIE of the Hessian diagonal at the precision
                # the two Hessians agree to? For every position i with perm_a[i] != perm_b[i]: the relative gap between the
                # two candidate channels' diagonal entries (arm a's values) against the largest relative difference of the
                # two arms' diagonals. gptq.py:58-64 sorts the fp32 diagonal: any gap below the summation-order noise of
                # H is decided by that noise.
                da_, db_ = res[ka]['H_diag'].astype(np.float64), res[kb]['H_diag'].astype(np.float64)
                dnoise = float((np.abs(da_ - db_) / np.maximum(np.abs(db_), 1e-300)).max())
                pa_, pb_ = res[ka][v + '/perm'].astype(np.int64), res[kb][v + '/perm'].astype(np.int64)
                dif = pa_ != pb_
                if dif.any():
                    gap = np.abs(da_[pa_[dif]] - da_[pb_[dif]]) / np.maximum(np.abs(da_[pa_[dif]]), 1e-300)
                    m['perm_diff_gap_med'] = float(np.median(gap))
                    m['perm_diff_within_4x_noise'] = float((gap <= 4 * dnoise).mean())
                else:
                    m['perm_diff_gap_med'], m['perm_diff_within_4x_noise'] = 0.0, 1.0
                m['H_diag_rel_max'] = dnoise
                srep['pairs'].setdefault(v, {})[ka + ' vs ' + kb] = m
                lines.append(f'{ka + " vs " + kb:<28}{m["codes_equal"]:>12.5f}{m["scale_rel_med"]:>15.3g}{m["scale_rel_max"]:>15.3g}'
                             f'{m["zeros_equal"]:>12.5f}{m["perm_equal"]:>11.4f}{m["w_rel_med"]:>11.3g}{m["w_rel_max"]:>11.3g}{m["H_rel_max"]:>11.3g}')
                lines.append(f'{"":<28}diag(H) rel diff max {m["H_diag_rel_max"]:.3g}; where the permutations differ: median diagonal gap '
                             f'{m["perm_diff_gap_med"]:.3g}, {100 * m["perm_diff_within_4x_noise"]:.1f} % of them within 4x that noise (ties)')
                # scales within 1e-4 / 1e-2 (north_star's tolerance is met by the bulk, not by the maximum: dynamic-group scales
                # are min / max of error-compensated weights)
                m['scales_within_1e-4'] = float((rel_s <= 1e-4).float().mean())
                m['scales_within_1e-2'] = float((rel_s <= 1e-2).float().mean())
                lines.append(f'{"":<28}scales within 1e-4: {m["scales_within_1e-4"]:.5f}, within 1e-2: {m["scales_within_1e-2"]:.5f}')
        report['shapes'].append(srep)
        del W, X, Xs, Y0
        torch.cuda.empty_cache()
    txt = '\n'.join(lines)
    print(txt)
    with open(a.out + '.json', 'w') as f:
        json.dump(report, f, indent=1)
    with open(a.out + '.txt', 'w') as f:
        f.write(txt + '\n')


if __name__ == '__main__':
    main()
