"""Recover the candidate weights k_clip_search builds (one-hot activations: err = (q_w[k] - w[k])^2) and compare them with
the oracle's, for the golden rows whose error entries differ. GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from llmc_amd.compression.quantization import IntegerQuantizer, awq_ops   # noqa: E402
from oracle import quant_ref as Q   # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'clip.npz'))
for name, row, gi in (('f16_asym_g128_noclipsym', 5, 0), ('f16_sym_g64_clipsym', 5, 2)):
    p = name + '/'
    sym, gs, clip_sym, nst = [int(v) for v in g[p + 'meta']]
    dt = str(g[p + 'dt'])
    tdt = torch.float16 if dt == 'f16' else torch.bfloat16
    W = g[p + 'w']
    wrow = W[row, gi * gs:(gi + 1) * gs].copy()
    q = IntegerQuantizer(4, bool(sym), 'per_group', group_size=gs)
    w = torch.from_numpy(wrow[None]).to(tdt).cuda()                      # [1, gs]
    qmin, qmax = Q.int_range(4, bool(sym))
    rec = np.zeros((10, gs), np.float32)
    for k in range(gs):
        x = torch.zeros(1, gs, dtype=tdt, device='cuda')
        x[0, k] = 1.0
        e = awq_ops.clip_errs(w, x, q, bool(clip_sym)).float().cpu().numpy()[:, 0, 0]     # (q_w[k] - w[k])^2 rounded
        rec[:, k] = e
    org_max = np.abs(wrow).max() if clip_sym else wrow.max()
    org_min = wrow.min()
    for i_s in range(10):
        f = np.float32(1 - i_s / 20)
        mx = Q.rnd(np.float32(org_max * f), dt)
        mn = -mx if clip_sym else Q.rnd(np.float32(org_min * f), dt)
        cw = np.minimum(np.maximum(wrow, mn), mx)
        qw, _, _ = Q.fake_quant_dynamic(cw.reshape(1, gs), dt, bool(sym), qmin, qmax)
        d = Q.rnd(qw.reshape(gs) - wrow, dt)
        want = Q.rnd(d * d, dt)
        bad = np.nonzero(want != rec[i_s])[0]
        print(name, 'row', row, 'group', gi, 'level', i_s, 'mx', float(mx), 'mn', float(mn), 'mismatching k:', bad[:8].tolist(),
              [(float(wrow[k]), float(qw.reshape(gs)[k]), float(want[k]), float(rec[i_s][k])) for k in bad[:3]])
