"""Where does k_clip_search differ from the oracle (which reproduces the reference's clip levels 100 %)? Compares the
kernel's error table (llmc_awq_clip_errs) with the oracle's, per golden case. GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from llmc_amd.compression.quantization import IntegerQuantizer, awq_ops   # noqa: E402
from oracle import awq_ref as A   # noqa: E402
from oracle import quant_ref as Q   # noqa: E402

TD = {'f16': torch.float16, 'bf16': torch.bfloat16}
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'clip.npz'))
for name in [str(n) for n in g['names']]:
    p = name + '/'
    sym, gs, clip_sym, nst = [int(v) for v in g[p + 'meta']]
    dt = str(g[p + 'dt'])
    x = g[p + 'x'].reshape(-1, g[p + 'x'].shape[-1])
    step = max(1, x.shape[0] // nst)
    xs = x[0::step]
    q = IntegerQuantizer(4, bool(sym), 'per_group', group_size=gs)
    w = torch.from_numpy(g[p + 'w']).to(TD[dt]).cuda()
    xd = torch.from_numpy(np.ascontiguousarray(xs)).to(TD[dt]).cuda()
    errs = awq_ops.clip_errs(w, xd, q, bool(clip_sym)).float().cpu().numpy()          # [ns, R, ng]
    eo = []
    qmin, qmax = Q.int_range(4, bool(sym))
    A.auto_clip_layer(g[p + 'w'], xs, dt, bool(sym), qmin, qmax, gs, bool(clip_sym), n_sample_token=xs.shape[0], errs_out=eo)
    eo = np.stack(eo)
    same = (errs == eo) | (np.isnan(errs) & np.isnan(eo))
    print(name, 'errs equal', same.mean(), 'per shrink level', [round(float(same[i].mean()), 4) for i in range(same.shape[0])])
    bad = np.argwhere(~same)
    for b in bad[:6]:
        i, r, gi = b
        print('   shrink', i, 'row', r, 'group', gi, 'kernel', errs[i, r, gi], 'oracle', eo[i, r, gi], 'rel', (errs[i, r, gi] - eo[i, r, gi]) / max(1e-30, abs(eo[i, r, gi])))
