import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
from conftest import load_golden
from llmc_amd.compression.quantization import IntegerQuantizer
g = load_golden('mse'); tot = same = 0
TD = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}
for ci, c in enumerate(g['cases']):
    dt, bit, sym, gran, gs = str(c).split('|'); sym = sym == 'True'; gs = None if gs == 'None' else int(gs)
    q = IntegerQuantizer(int(bit), sym, gran, calib_algo='mse', **(dict(group_size=gs) if gs else {}))
    w = torch.from_numpy(g[f'c{ci}_w']).to(TD[dt]).cuda()
    mn, mx = q.get_tensor_range(q.reshape_tensor(w))
    s = (mn.float().cpu().numpy().reshape(-1) == g[f'c{ci}_min']) & (mx.float().cpu().numpy().reshape(-1) == g[f'c{ci}_max'])
    tot += s.size; same += int(s.sum())
print('rows identical to the reference:', same, 'of', tot)
q = IntegerQuantizer(4, False, 'per_group', calib_algo='mse', group_size=128)
w = (torch.randn(4096, 4096, device='cuda') * 0.02).to(torch.bfloat16)
q.fake_quant_weight_dynamic(w); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); q.fake_quant_weight_dynamic(w); e1.record(); torch.cuda.synchronize()
print('mse W4 g128 fake-quant of a 4096x4096 bf16 weight: %.2f ms' % e0.elapsed_time(e1))
